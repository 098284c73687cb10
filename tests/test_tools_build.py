"""The standalone microbenchmarks under tools/microbench/ must keep compiling for gfx950 (they are run by hand on the GPU box;
``hipcc`` cross-compiles without a GPU).  No kernel is launched here."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SOURCES = sorted(glob.glob(os.path.join(ROOT, "tools", "microbench", "*.hip")))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", SOURCES, ids=[os.path.basename(s) for s in SOURCES])
def test_microbenchmark_compiles_for_gfx950(src, tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-o", str(tmp_path / "a.out"), src], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert os.path.getsize(tmp_path / "a.out") > 0


PY_TOOLS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))


@pytest.mark.parametrize("src", PY_TOOLS, ids=[os.path.basename(s) for s in PY_TOOLS])
def test_tool_script_compiles(src):
    """the measurement / probe scripts under tools/ (run by hand on the GPU box) must at least stay valid Python."""
    import py_compile
    py_compile.compile(src, doraise=True)
