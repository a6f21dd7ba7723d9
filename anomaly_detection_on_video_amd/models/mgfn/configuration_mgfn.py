"""Hyper-parameters of the MGFN scorer; same 14 keyword arguments and defaults as the reference
(`/root/reference/src/models/mgfn/configuration_mgfn.py:4-36`, `configs/runner/mgfn.yaml:5-19`).

Subclasses `transformers.PretrainedConfig` when transformers is importable (so `save_pretrained`
/ `from_pretrained` keep working for users of the reference), else a plain attribute bag.
"""
try:  # transformers is plumbing here, not a hard dependency
    from transformers.configuration_utils import PretrainedConfig as _Base
except Exception:  # pragma: no cover
    class _Base:  # type: ignore
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                setattr(self, k, v)


class MGFNConfig(_Base):
    model_type = "mgfn"

    def __init__(
        self,
        classes=0,
        dims=(64, 128, 1024),
        depths=(3, 3, 2),
        mgfn_types=("gb", "fb", "fb"),
        lokernel=5,
        channels=2048,
        ff_repe=4,
        dim_head=64,
        local_aggr_kernel=5,
        dropout=0.0,
        attention_dropout=0.0,
        dropout_rate=0.7,
        mag_ratio=0.1,
        k=3,
        **kwargs,
    ):
        super().__init__(**kwargs)
        self.classes = classes
        self.dims = tuple(dims)
        self.depths = tuple(depths)
        self.mgfn_types = tuple(mgfn_types)
        self.lokernel = lokernel
        self.channels = channels
        self.ff_repe = ff_repe
        self.dim_head = dim_head
        self.local_aggr_kernel = local_aggr_kernel
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.dropout_rate = dropout_rate
        self.mag_ratio = mag_ratio
        self.k = k
        if not (len(self.dims) == len(self.depths) == len(self.mgfn_types)):
            raise ValueError("dims, depths and mgfn_types must have one entry per stage")
