"""CPU: structure of the restated MONAI-0.3 UNet and the tiny-step fixture."""
import numpy as np
import torch

from oracle.monai_unet import UNet
from oracle.trainer import OracleUNet3D


def test_config_b_parameter_count_and_keys():
    net = UNet(3, 1, 10, (32, 64, 128, 256), (2, 2, 2, 2), num_res_units=2)
    assert sum(p.numel() for p in net.parameters()) == 4_756_301
    keys = set(net.state_dict())
    for k in ("model.0.conv.unit0.conv.weight", "model.0.conv.unit0.act.weight", "model.0.residual.bias",
              "model.1.submodule.1.submodule.1.submodule.conv.unit1.conv.weight",
              "model.1.submodule.1.submodule.1.submodule.residual.weight",
              "model.2.0.conv.weight", "model.2.0.act.weight", "model.2.1.conv.unit0.conv.bias"):
        assert k in keys, k
    assert "model.2.1.conv.unit0.act.weight" not in keys  # last_conv_only at the top
    # the reference's own indexing of the tree (capstone/interpretability.py:88)
    assert isinstance(net.model[2][1].conv.unit0.conv, torch.nn.Conv3d)
    assert net.model[2][0].conv.weight.shape == (64, 10, 3, 3, 3)  # ConvTranspose: (Cin, Cout, k,k,k)
    assert net.model[1].submodule[1].submodule[1].submodule.residual.kernel_size == (1, 1, 1)


def test_five_filter_variant_and_2d():
    net = UNet(3, 1, 10, (32, 64, 128, 256, 512), (2, 2, 2, 2), num_res_units=2)
    assert sum(p.numel() for p in net.parameters()) == 19_233_361
    y = UNet(2, 1, 10, (4, 8, 16, 32, 64), (2, 2, 2, 2), num_res_units=0)(torch.zeros(1, 1, 32, 32))
    assert y.shape == (1, 10, 32, 32)


def test_parameter_counts_the_reference_report_publishes():
    """The only reference-held numbers that touch the (otherwise unpinned) MONAI topology: reports/Report.pdf p.8 Table 1 —
    Model L (2-D U-Net, filters 64..1024, 2 residual units) has 26 M parameters, Model M (1 residual unit per the Report;
    the 2-D CLI's use_res_units switch gives 2 or 0) 13.5 M.  Built exactly as capstone/training/base_trainer.py:61-79 builds
    them: UNet(dimensions=2, in_channels=1, out_channels=10, channels=filters, strides=[2,2,2,2], num_res_units=...)."""
    filters = (64, 128, 256, 512, 1024)
    count = lambda nres: sum(p.numel() for p in UNet(2, 1, 10, filters, (2, 2, 2, 2), num_res_units=nres).parameters())
    n_l, n_m = count(2), count(1)
    assert n_l == 25_980_905 and round(n_l / 1e6) == 26            # "26 M"
    assert n_m == 13_408_292 and abs(n_m / 1e6 - 13.5) < 0.1       # "13.5 M"


def test_tiny_step_fixture_reproduces(golden):
    g = golden("unet_tiny.npz")
    for tag in ("a", "b"):
        m = OracleUNet3D(filters=tuple(g[f"{tag}_filters"]), loss_fx=tuple(str(s) for s in g[f"{tag}_losses"]))
        m.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_w:")})
        batch = tuple(torch.from_numpy(g[f"{tag}_{n}"]) for n in ("images", "masks", "indicator"))
        opt = m.configure_optimizers()
        loss = m.fit_step(batch, opt)
        np.testing.assert_allclose(loss.numpy(), g[f"{tag}_loss"], rtol=1e-5)
        np.testing.assert_allclose(m.logged["Mean Dice Score (train)"].numpy(), g[f"{tag}_dice_mean"], atol=1e-6)
        for k, v in m.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f"{tag}_w1:{k}"], rtol=1e-4, atol=2e-4)
