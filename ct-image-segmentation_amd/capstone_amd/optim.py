"""``configure_optimizers()``'s optimizer (reference capstone/volumetric/base_trainer.py:113-114 and
capstone/training/base_trainer.py:138-148: ``optim.Adam(self.parameters(), lr=self.hparams.lr)``).

``Adam`` IS a ``torch.optim.Adam`` (same constructor, ``param_groups``, ``state_dict()`` layout, scheduler / Lightning
interplay); its ``step()`` runs the U-Net's parameters — views of one flat fp32 buffer whose gradients the backward kernels
wrote into the matching flat gradient buffer — through ONE ``ctseg_adam_step`` launch instead of torch's per-tensor foreach
kernels, and rebuilds the packed MFMA operands on the side stream as ``fit_step`` does.  Parameters that are not the engine's
(the 2-D module's ``conv1x1``) take torch's own step.  Anything the kernel does not implement (weight decay, amsgrad,
maximize, several differing parameter groups) falls back to ``torch.optim.Adam.step`` for everything, decided at construction.
"""
import torch


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, unet=None, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._unet = unet
        g = self.param_groups
        self._native = (unet is not None and len(g) == 1 and g[0]["weight_decay"] == 0 and not g[0]["amsgrad"]
                        and not g[0].get("maximize", False) and not g[0].get("capturable", False)
                        and not g[0].get("differentiable", False))

    # ---- which parameters are the engine's -----------------------------------------------------------------------
    def _store(self, ensure=False):
        if not self._native:
            return None
        eng = self._unet.engine()
        if ensure:
            # no flat store yet (a fresh module whose optimizer state is loaded before its first forward — torch's usual resume
            # order), or the Parameters were re-homed since: build / re-attach it where the parameters live now
            own = next(iter(self._unet.parameters()), None)
            if own is not None:
                eng.ensure(own.device)
        st = eng.store
        if st is None or not st.attached():
            return None
        mine = {id(p) for p in self.param_groups[0]["params"]}
        return st if all(id(p) in mine for p in st.params) else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = self._store()
        if st is None:
            super().step()
            return loss
        grads = [p.grad for p in st.params]
        if all(g is None for g in grads):
            super().step()                     # nothing was differentiated: torch's step skips every parameter too
            return loss
        if any(g is None for g in grads):
            raise RuntimeError("some of the U-Net's parameters have no gradient: the native Adam step updates the flat buffer as a whole")
        base = st.flat_g.data_ptr()
        for p, g in zip(st.params, grads):     # a gradient that is not the flat buffer's own view (clipped copy, hook output)
            if g.data_ptr() != base + 4 * st.off(p):
                st.grad_view(p).copy_(g)
        grp = self.param_groups[0]
        st.adam_step(float(grp["lr"]), tuple(grp["betas"]), float(grp["eps"]))
        eng = self._unet.engine()
        plan = eng.last_plan
        if plan is not None and not plan.inference and plan.store is st:
            plan.repack_after_update()
        own = {id(p) for p in st.params}
        foreign = [p for p in grp["params"] if id(p) not in own and p.grad is not None]
        if foreign:                            # e.g. BaseUNet2D.conv1x1: torch's step on those alone
            for p in st.params:
                p.grad = None
            try:
                super().step()
            finally:
                for p, g in zip(st.params, grads):
                    p.grad = g
        return loss

    # ---- torch.optim.Adam's state_dict layout, the engine's moments included ---------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        st = self._store()
        if st is not None and st.step > 0 and st.adam_m is not None:
            index = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
            for p in st.params:
                o, n = st.off(p), p.numel()
                sd["state"][index[id(p)]] = {"step": torch.tensor(float(st.step)),
                                             "exp_avg": st.adam_m[o:o + n].view(p.shape).clone(),
                                             "exp_avg_sq": st.adam_v[o:o + n].view(p.shape).clone()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self._store(ensure=True)
        if st is None:
            return
        have = [p for p in st.params if p in self.state and "exp_avg" in self.state[p]]
        if not have:
            return
        if len(have) != len(st.params):
            raise ValueError("optimizer state covers only part of the U-Net's parameters")
        steps = {int(float(self.state[p]["step"])) for p in have}
        if len(steps) != 1:
            raise ValueError("the U-Net's parameters carry different step counts")
        st.ensure_adam_state()
        for p in have:
            o, n = st.off(p), p.numel()
            e = self.state.pop(p)
            st.adam_m[o:o + n].copy_(e["exp_avg"].reshape(-1))
            st.adam_v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
        st.step = steps.pop()
