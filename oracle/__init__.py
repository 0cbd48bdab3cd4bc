"""CPU oracle for the 3D U-Net hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker. The product path
(``ct-image-segmentation_amd/``) never imports it and has no CPU fallback.

Parity pin status (see DESIGN.md "Oracle"):
  * ``oracle.metrics`` / ``oracle.losses`` (squash, softmax-argmax, mean-Dice,
    metric reduction, CE wrappers, missing-annotation weighting, GDL) are pinned
    against fixtures produced by *running the reference's own leaf files*
    (``tests/golden/make_golden.py``).
  * ``oracle.monai_unet`` (MONAI 0.3 ``UNet`` topology), MONAI ``DiceLoss`` and
    ``FocalLoss`` restate a third-party dependency (``monai==0.3``,
    reference README.md:39) that is absent from /root/reference and from this
    image: **parity unpinned** for those; the primitive arithmetic underneath is
    plain ``torch.nn`` on CPU, which is what MONAI itself composes.
"""
