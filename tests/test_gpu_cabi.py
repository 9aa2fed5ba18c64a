"""A plain C program against include/rdx.h + librdx.so (no Python, no torch in the process)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_the_library(tmp_path):
    exe = str(tmp_path / "smoke")
    lib_dir = os.path.join(ROOT, "rag_dpo_amd")
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "smoke.c"),
                           "-L", lib_dir, "-l:librdx.so", "-lm", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c-abi smoke ok" in out.stdout


def test_c_abi_argument_checks_with_a_device(tmp_path):
    """tests/c_abi/errors.c under ASan + UBSan with a GPU present: the handle-level checks run too"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("test_sanitizers", os.path.join(ROOT, "tests", "test_sanitizers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = m.build_and_run_errors(tmp_path)
    assert "(with a device)" in out
