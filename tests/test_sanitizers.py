"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU side (GPU sanitizers do not exist on this pool):
the C oracle on inputs that reach every loop bound it has, and the C-ABI's argument checks called from a plain C program."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _run(cmd, env=ENV, timeout=300):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_oracle_under_asan_and_ubsan(tmp_path):
    src = [os.path.join(ROOT, "tests", "c_abi", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "rdx_oracle.c")]
    flags = ["-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-Wall", "-Wextra"]
    san, plain = str(tmp_path / "oracle_san"), str(tmp_path / "oracle_plain")
    subprocess.check_call(["gcc", *SAN, *flags, *src, "-lm", "-o", san])
    subprocess.check_call(["gcc", "-O3", "-march=native", *flags, *src, "-lm", "-o", plain])
    a = re.search(r"checksum ([0-9a-f]{16})", _run([san], env=dict(ENV, OMP_NUM_THREADS="3")))
    b = re.search(r"checksum ([0-9a-f]{16})", _run([plain]))
    assert a and b and a.group(1) == b.group(1)      # same results instrumented at -O1 and as shipped (-O3 -march=native)


def build_and_run_errors(tmp_path):
    lib_dir = os.path.join(ROOT, "rag_dpo_amd")
    assert os.path.exists(os.path.join(lib_dir, "librdx.so")), "build the library first (python -m rag_dpo_amd.build)"
    exe = str(tmp_path / "errors")
    subprocess.check_call(["gcc", *SAN, "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "errors.c"),
                           "-L", lib_dir, "-l:librdx.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    # (leak detection off here: the HIP runtime librdx links keeps allocations of its own until process exit)
    out = _run([exe], env=dict(ENV, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1"))
    assert "c-abi error paths ok" in out
    return out


def test_c_abi_argument_checks_under_asan_and_ubsan(tmp_path):
    build_and_run_errors(tmp_path)
