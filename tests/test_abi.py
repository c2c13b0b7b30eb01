"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ov2slam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ov2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ov2slam_amd
    from ov2slam_amd import _lib
    lib = ov2slam_amd.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libov2slam_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.ov2_version() == _lib.OV2_ABI_VERSION == 600


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    import ov2slam_amd
    try:
        ctx = ov2slam_amd.Context(0)
    except ov2slam_amd.Ov2Error as e:
        assert e.code == -5          # OV2_ENODEVICE
        return
    ctx.close()                       # a GPU is present: nothing to check here


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "ov2slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "ov2_oracle.h" not in txt and "orc_" not in txt, os.path.join(dp, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_library_path_override_and_missing_library_fail_loudly(tmp_path):
    """OV2SLAM_HIP_LIB selects the build to load (A/B timing builds); a path without a library is an ImportError
    that names the build command -- never a silent CPU path."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import ov2slam_amd; from ov2slam_amd import _lib; "
            "print(_lib.LIB_PATH); ov2slam_amd.load(); print('loaded')" % ROOT)
    good = os.path.join(ROOT, "ov2slam_amd", "libov2slam_hip.so")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OV2SLAM_HIP_LIB=good), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split()[0] == good and "loaded" in r.stdout, r.stderr
    missing = str(tmp_path / "nope.so")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OV2SLAM_HIP_LIB=missing), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ImportError" in r.stderr and "no CPU fallback" in r.stderr.replace("\n", " "), r.stderr


def test_library_reads_the_environment_only_at_context_creation():
    """The entry points run on several threads of a host process that may call setenv concurrently (a ROS node): kernel-path
    switches are per-context options (ov2_ctx_set_option), and the only getenv of the product sits in ov2_ctx_create (OV2_DEBUG)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = []
    for f in sorted(glob.glob(os.path.join(root, "ov2slam_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "ov2slam_amd", "host", "*.hpp"))):
        for i, line in enumerate(open(f), 1):
            if re.search(r"\bgetenv\s*\(", line):
                hits.append((os.path.basename(f), i))
    assert [h[0] for h in hits] == ["ctx.hip"], hits
    assert "__CUDACC__" not in open(os.path.join(root, "ov2slam_amd", "csrc", "xcd_map.hpp")).read()
