"""Resolve the reference's OWN import name for the hot path -- ``model.modeling.build_model`` (/root/reference/train.py:30, test.py:21) --
to this build, without editing or copying a file of the reference tree:

    import csbsr_amd.dropin; csbsr_amd.dropin.install()        # first line of train.py / test.py, or in sitecustomize
    from model.modeling.build_model import JointModelWithLoss, JointModel          # unchanged reference code: now csbsr_amd's classes

``install()`` puts ONE finder in front of ``sys.meta_path`` that answers exactly that module name; every other ``model.*`` import
(``model.config``, ``model.engine.trainer``, ``model.data...``) keeps coming from the reference tree on ``sys.path``.  Where the reference
tree is absent (this repo's tests, the GPU box) the two parent packages are created empty so the import statement itself still works.
The names of that module this build does not implement (``JointInvModelWithLoss``, ``SRModelWithLoss``, ``JointInvModel``: MODEL.SR_SEG_INV
/ JOINT_LEARNING = False, outside SURVEY section 8) are present and raise ``NotImplementedError`` when constructed -- the reference's own
behaviour for a model name it does not know (build_model.py:114,244,315), never a silent fallback to its torch path.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types

TARGET = "model.modeling.build_model"
_UNBUILT = ("JointInvModelWithLoss", "SRModelWithLoss", "JointInvModel")


def _unbuilt(name):
    class _Unbuilt:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"csbsr_amd does not build {name} (SR_SEG_INV / JOINT_LEARNING=False are outside the hot path: "
                                      f"SURVEY.md section 8); use the reference's own model.modeling.build_model for it")
    _Unbuilt.__name__ = _Unbuilt.__qualname__ = name
    return _Unbuilt


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == TARGET:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return types.ModuleType(spec.name)

    def exec_module(self, module):
        from csbsr_amd.modeling import build_model as B
        module.JointModelWithLoss, module.JointModel = B.JointModelWithLoss, B.JointModel
        for n in _UNBUILT:
            setattr(module, n, _unbuilt(n))
        module.__doc__ = "csbsr_amd.dropin: the reference's model.modeling.build_model bound to csbsr_amd.modeling.build_model"


def _ensure_parent(name):
    """make ``name`` importable as a package: the reference's own if it is on sys.path, else an empty placeholder"""
    if name in sys.modules:
        return
    try:
        found = importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        found = False
    if not found:
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
        if "." in name:
            setattr(sys.modules[name.rsplit(".", 1)[0]], name.rsplit(".", 1)[1], pkg)


def install():
    """idempotent; returns True when the finder was added by this call"""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return False
    sys.modules.pop(TARGET, None)        # (an already imported reference module would win over any finder)
    _ensure_parent("model")
    _ensure_parent("model.modeling")
    sys.meta_path.insert(0, _Finder())
    return True


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _Finder)]
    sys.modules.pop(TARGET, None)
