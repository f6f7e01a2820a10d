from hiast_amd.utils.registry.registries import SEG_MODEL


def build_seg_model(cfg):
    """seg_models/__init__.py:5-8 of the reference"""
    kind = cfg.model.seg_model.type
    if kind != "DeepLab_V2":
        raise AssertionError("only DeepLab_V2 is available, got %r" % (kind,))
    return SEG_MODEL[kind](num_classes=cfg.dataset.num_classes, output_dim=cfg.model.seg_model.output_dim)
