"""Import modules of the reference (bupt-ai-cz/HIAST, mounted read-only at /root/reference)
in the BUILD CONTAINER ONLY, with stub modules for the dependencies the image lacks
(torchvision, cv2, albumentations, apex, tensorboardX, yacs, imageio).

Used by tests/golden/make_golden.py (fixture generation) and by the tests marked
`needs_reference`, which are skipped wherever /root/reference is absent (e.g. the GPU box).
Nothing from the reference is copied: its modules are imported and called.
"""
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/code"


def available():
    return os.path.isdir(REF_ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_captured_pngs = {}


def captured_pngs():
    """arrays handed to the stubbed cv2.imwrite, keyed by path"""
    return _captured_pngs


def install_stubs():
    import numpy as np
    if not hasattr(np, "bool"):
        np.bool = bool  # sseg/datasets/preprocessor.py:103 predates numpy 1.24
    if "numpy.lib.type_check" not in sys.modules:   # preprocessor.py:5 imports it (unused there)
        try:
            importlib.import_module("numpy.lib.type_check")
        except ImportError:
            _stub("numpy.lib.type_check", common_type=np.common_type)

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tvm = _stub("torchvision.models")
        tvu = _stub("torchvision.models.utils",
                    load_state_dict_from_url=lambda *a, **k: (_ for _ in ()).throw(
                        RuntimeError("no network")))
        tv.models = tvm
        tvm.utils = tvu
        tvt = _stub("torchvision.transforms")
        tv.transforms = tvt

        import torch

        class ToTensor:
            def __call__(self, img):
                a = np.asarray(img)
                t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
                return t.float().div(255)

        class Normalize:
            def __init__(self, mean, std):
                self.mean = torch.tensor(mean).view(-1, 1, 1)
                self.std = torch.tensor(std).view(-1, 1, 1)

            def __call__(self, t):
                return (t - self.mean) / self.std

        class Compose:
            def __init__(self, fs):
                self.fs = fs

            def __call__(self, x):
                for f in self.fs:
                    x = f(x)
                return x

        tvt.ToTensor, tvt.Normalize, tvt.Compose = ToTensor, Normalize, Compose

    if "cv2" not in sys.modules:
        def imwrite(path, arr):
            _captured_pngs[path] = np.array(arr)
            return True

        def resize(img, dsize, interpolation=0):
            # only the identity case is exercised by the goldens (same-size label maps)
            assert tuple(dsize) == (img.shape[1], img.shape[0]), "stub cv2.resize: identity only"
            return img

        ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)
        _stub("cv2", imwrite=imwrite, resize=resize, INTER_LINEAR=1, INTER_NEAREST=0,
              setNumThreads=lambda *_: None, ocl=ocl)

    if "albumentations" not in sys.modules:
        _stub("albumentations")
    if "apex" not in sys.modules:
        ap = _stub("apex")
        par = _stub("apex.parallel", SyncBatchNorm=type("SyncBatchNorm", (), {}),
                    convert_syncbn_model=lambda m: m,
                    DistributedDataParallel=object)
        amp = _stub("apex.amp")
        ap.parallel, ap.amp = par, amp
    if "tensorboardX" not in sys.modules:
        _stub("tensorboardX", SummaryWriter=object)
    if "imageio" not in sys.modules:
        _stub("imageio")
    if "tqdm" not in sys.modules:
        _stub("tqdm", tqdm=lambda it, **k: it)


def ref(module_name):
    """import `module_name` (e.g. 'sseg.models.modules.losses') from the reference tree"""
    assert available(), "/root/reference is not mounted here"
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module(module_name)


def ref_deeplab_v2(num_classes=19, output_dim=256):
    """Instantiate the reference DeepLab_V2 without the ImageNet download (deeplab_v2.py:33)."""
    dl = ref("sseg.models.modules.seg_models.deeplab_v2")
    orig = dl.build_resnet101
    dl.build_resnet101 = lambda pretrained=True, **k: orig(pretrained=False, **k)
    try:
        return dl.DeepLab_V2(num_classes=num_classes, output_dim=output_dim)
    finally:
        dl.build_resnet101 = orig
