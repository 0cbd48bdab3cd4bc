"""CPU only: the HOST half of libctseg_hip.so (descriptor validation, kernel selection, sizing queries, argument marshalling) under
AddressSanitizer + UndefinedBehaviorSanitizer (`make asan`: device code compiled as usual, host code instrumented).  SURVEY.md
section 5's race / sanitizer row; the GPU pool offers no device sanitizer, so this never runs on the GPU box (and lib_asan/ is in
.gpurunignore)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ct-image-segmentation_amd")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asan_rt():
    out = subprocess.run(["make", "-s", "-C", PKG, "asan_rt"], capture_output=True, text=True).stdout.strip()
    return out if out and os.path.exists(out) else None


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc missing")
def test_host_code_is_clean_under_asan_and_ubsan():
    import torch
    if torch.cuda.is_available():
        pytest.skip("host-sanitizer build is for the CPU container only")
    rt = _asan_rt()
    if rt is None:
        pytest.skip("clang AddressSanitizer runtime not found")
    subprocess.check_call(["make", "-C", PKG, "-j8", "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(PKG, "lib_asan", "libctseg_hip.so")
    env = dict(os.environ, LD_PRELOAD=rt, CTSEG_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=67")
    # (1) the random-descriptor driver
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitizer_child.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "SANITIZER-CHILD-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    # (2) the ABI export / validation tests against the instrumented library
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_exports.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
