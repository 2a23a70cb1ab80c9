"""
ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Recipe for `oracle/_ref/`: the UNMODIFIED reference optimizer, staged where it can travel to the GPU box.

    python oracle/make_ref.py            (also called by __graft_entry__.build())

The reference is pure Python: its "binary" is the module itself.  `/root/reference` exists only in the authoring
container, so the one module of the hot path -- tangram/mapping_optimizer.py (imports numpy, logging, torch only; `import
tangram` as a package needs scanpy, which is absent) -- is copied byte for byte into `oracle/_ref/`, which is listed in
.gitignore (never in the history) and not in .gpurunignore (ships with the snapshot, like the built .so files).  A manifest
records the source path and the SHA-256 of the copy, `load()` verifies it before every use.

Who may use it: `tests/` (the live-reference GPU test), `bench.py`'s cpu_baseline leg (`"kind": "reference"`), nothing else.
"""
import hashlib
import importlib.util
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/tangram/mapping_optimizer.py"
REF_DIR = os.path.join(HERE, "_ref")
REF_COPY = os.path.join(REF_DIR, "ref_mapping_optimizer.py")
MANIFEST = os.path.join(REF_DIR, "MANIFEST.json")


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def make(verbose=False):
    """Stage the reference module.  Returns the path of the copy, or None when the reference checkout is absent (GPU box:
    the copy made in the authoring container is used as it is)."""
    if not os.path.exists(REF_SRC):
        return REF_COPY if available() else None
    os.makedirs(REF_DIR, exist_ok=True)
    shutil.copyfile(REF_SRC, REF_COPY)
    json.dump({"source": REF_SRC, "sha256": _sha(REF_COPY), "bytes": os.path.getsize(REF_COPY),
               "note": "byte-for-byte copy of the reference's hot-path module; git-ignored, staged by oracle/make_ref.py"},
              open(MANIFEST, "w"), indent=1)
    if verbose:
        print("staged", REF_COPY, _sha(REF_COPY)[:16])
    return REF_COPY


def available():
    if not (os.path.exists(REF_COPY) and os.path.exists(MANIFEST)):
        return False
    try:
        return json.load(open(MANIFEST))["sha256"] == _sha(REF_COPY)
    except Exception:
        return False


def load():
    """The reference module (`Mapper`, `MapperConstrained`), loaded standalone from the staged copy."""
    if not available():
        raise RuntimeError("oracle/_ref is not staged (run `python oracle/make_ref.py` where /root/reference exists)")
    name = "tangram_reference_mapping_optimizer"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, REF_COPY)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = make(verbose=True)
    print(p if p else "reference checkout not present and no staged copy")
