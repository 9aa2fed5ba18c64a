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


def test_sharded_overflow_rehearsal_repeats_the_exchange_on_every_rank():
    """the repeated-exchange branch of the sharded search with three ranks: `bench.py --gpus 3 --set cand_cap=8` makes
    EVERY search overflow its candidate segments, so every rank re-runs queries after its kernels and all ranks must learn it from
    the merged flags word and exchange a second time. Three processes share this box's one GPU and the collective runs over gloo
    through host memory (RDX_BENCH_REHEARSAL=1: the rehearsal form; the packed layout, the flags word, the merge kernel and the
    rdx_signal wait are the real ones). Asserts: two exchanges per search, the same merged checksum on all ranks, and the merged
    lists equal to the per-shard exact scans merged the same way."""
    import json
    import sys
    env = dict(os.environ, RDX_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--workload", "c3", "--rows", "300000", "--steps", "3",
                          "--warmup", "1", "--no-cpu", "--set", "cand_cap=8"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 3 and d["rccl"]["world"] == 3
    chk, bd = d["distributed_check"], d["step_breakdown"]
    assert chk["merged_identical_on_all_ranks"] is True and chk["per_shard_exact_scan_merged_equals_mfma_merged"] is True, chk
    assert bd["exchanges"] == 2 * bd["steps"], bd                   # every search: the first exchange, then the repeated one
    assert d["path_stats"]["retried_queries"] > 0 or d["path_stats"]["exact_fallback_queries"] > 0
