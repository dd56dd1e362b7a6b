"""Drop-in boundary check: every name the reference's run / eval scripts import from the package,
every function of ours they call and every keyword they pass must resolve against THIS package.

The reference scripts are only PARSED (ast) where they lie under /root/reference -- nothing of them is imported,
executed or copied; on a box without the reference checkout (the GPU box) the test skips."""
import ast
import dataclasses
import glob
import importlib
import inspect
import os

import pytest

REF = "/root/reference"
# the experiment / evaluation scripts (the callers of the surface); runs/_model.py, _helper.py and _loader.py ARE the
# surface and are replaced by this package's own files of the same names
SCRIPTS = sorted(p for p in glob.glob(os.path.join(REF, "runs", "*.py")) + glob.glob(os.path.join(REF, "runs", "eval", "*.py"))
                 if not os.path.basename(p).startswith("_"))
OURS = ("future_od", "runs", "config")

pytestmark = pytest.mark.skipif(not SCRIPTS, reason="reference checkout not present")


def _accepts(fn, kw):
    if dataclasses.is_dataclass(fn):
        return kw in {f.name for f in dataclasses.fields(fn)}
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return True
    return kw in sig.parameters or any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, REF) for p in SCRIPTS])
def test_reference_script_resolves_against_the_package(path):
    tree = ast.parse(open(path).read())
    bound = {}                                      # local name -> object of ours
    checked = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in OURS:
            mod = importlib.import_module(node.module)
            for a in node.names:
                if not hasattr(mod, a.name):            # a submodule: from future_od.datasets import nu_scenes
                    try:
                        importlib.import_module(f"{node.module}.{a.name}")
                    except ImportError:
                        pass
                assert hasattr(mod, a.name), f"{os.path.basename(path)}: {node.module} has no {a.name}"
                bound[a.asname or a.name] = getattr(mod, a.name)
                checked += 1
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in OURS:
                    mod = importlib.import_module(a.name)
                    bound[a.asname or a.name.split(".")[0]] = mod if a.asname else importlib.import_module(a.name.split(".")[0])
                    checked += 1
    for node in ast.walk(tree):
        # attribute of an imported module / class: nu_scenes.CATEGORY_DICT, T.JointResize, transformer.MLP ...
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in bound:
            obj = bound[node.value.id]
            if inspect.ismodule(obj) or inspect.isclass(obj):
                assert hasattr(obj, node.attr), f"{os.path.basename(path)}: {node.value.id}.{node.attr} missing"
                checked += 1
        # keywords of calls to imported callables
        if isinstance(node, ast.Call):
            fn = None
            if isinstance(node.func, ast.Name) and node.func.id in bound:
                fn = bound[node.func.id]
            elif (isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name)
                  and node.func.value.id in bound and hasattr(bound[node.func.value.id], node.func.attr)):
                fn = getattr(bound[node.func.value.id], node.func.attr)
            if fn is None or not callable(fn):
                continue
            for kw in node.keywords:
                if kw.arg is not None:
                    assert _accepts(fn, kw.arg), f"{os.path.basename(path)}: {getattr(fn, '__name__', fn)}() takes no '{kw.arg}'"
                    checked += 1
            if not dataclasses.is_dataclass(fn) and not inspect.isclass(fn):
                try:
                    params = [p for p in inspect.signature(fn).parameters.values()
                              if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
                    varargs = any(p.kind == p.VAR_POSITIONAL for p in inspect.signature(fn).parameters.values())
                    assert varargs or len(node.args) <= len(params), f"{os.path.basename(path)}: too many positionals for {fn.__name__}"
                except (TypeError, ValueError):
                    pass
    assert checked > 0 or os.path.basename(path) == "helpers.py"        # (helpers.py imports nothing of ours)


def test_trainer_surface_used_by_the_scripts():
    from future_od.trainer import Trainer
    for name in ("train", "eval", "load_checkpoint", "save_checkpoint"):
        assert hasattr(Trainer, name), name
    from future_od.datasets import nu_images, nu_scenes
    assert len(nu_scenes.CATEGORY_DICT) == 8 and nu_images.ANNOTATED_FRAME == 6
    from config import config
    assert {"visualization_path", "checkpoint_path", "nuscenes_path", "nuimages_path"} <= set(config)


def test_positional_order_of_the_loader_factories_matches_the_reference():
    from runs._loader import get_nuim_loaders, get_nusc_loaders
    for fn in (get_nusc_loaders, get_nuim_loaders):
        names = list(inspect.signature(fn).parameters)[:7]
        assert names == ["img_size", "offsets", "args", "config", "train_batch_size", "random_aug",
                         "val_annotated_frame_override"], names
