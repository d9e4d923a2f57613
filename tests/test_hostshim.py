"""The product's HOST logic without a GPU, under sanitizers.

tests/hostshim/ builds the host C sources of the product (clownresampler_amd/csrc: cr_api.c, cr_context.c, cr_plan.c, cr_device_api.c,
cr_multi.c - batching, ticket-block rings, plan cache, helper threads, streaming windows, callback replay, segments, the sharded call)
against tests/hostshim/crhip_fake.c, a plain-C implementation of the HIP seam csrc/crhip.h whose "kernels" are scalar models over the
launch arguments (test infrastructure: it calls oracle/, it is never linked into libclownresampler_amd.so), and a C driver that holds every
result to the oracle.  Run here: plain, -fsanitize=address,undefined (leak detection on) and -fsanitize=thread."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "hostshim")
BUILD = os.path.join(SHIM, "build")


@pytest.fixture(scope="module")
def drivers():
    targets = [os.path.join(BUILD, n) for n in ("driver", "driver_asan", "driver_tsan")]
    r = subprocess.run(["make", "-s", "-j4", "-C", SHIM] + targets, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return targets


def _run(exe, env_extra, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=env)
    tail = (r.stdout + r.stderr)[-6000:]
    assert r.returncode == 0 and "driver: all passed" in r.stdout, tail
    for name in ("bulk", "bulk_batches", "callback", "stream", "concurrent", "segments", "sharded", "device_threads", "failures", "shutdown"):
        assert "ok " + name in r.stdout, tail
    return r


def test_host_logic_against_the_oracle(drivers):
    _run(drivers[0], {}, 300)


def test_host_logic_under_address_and_ub_sanitizers(drivers):
    r = _run(drivers[1], {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}, 600)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]


def test_host_logic_under_thread_sanitizer(drivers):
    r = _run(drivers[2], {"TSAN_OPTIONS": "halt_on_error=0:exitcode=66"}, 900)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]


def test_fake_seam_is_not_in_the_product():
    """the stand-in is test infrastructure: the shipped library exports nothing of it and does not contain the oracle"""
    import clownresampler_amd as cr
    out = subprocess.run(["nm", "-D", "--defined-only", cr.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "crhip_fake" not in out and "oracle_" not in out
    out = subprocess.run(["nm", cr.LIB_PATH], capture_output=True, text=True).stdout
    assert "crhip_fake_launches" not in out and "oracle_frame" not in out


def test_soak_through_the_python_mirror_under_address_sanitizer(drivers):
    """tests/soak_gpu.py (the randomised soak the GPU box runs against the real kernels) pointed at the sanitizer build of the fake-seam
    library through the ctypes mirror: every entry point's HOST logic - timeline walk, 4 Mi batching, states, callbacks, streaming
    windows, scripted high-level sessions with Adjusts - on random configurations, under AddressSanitizer, a minute of it.  (Round 5:
    this is how the high-level flush's pull schedule and the too-wide staging window were debugged without a GPU.)"""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ)
    env.update({"CLOWNRESAMPLER_AMD_LIBRARY": os.path.join(BUILD, "libcr_hostshim_asan.so"), "LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0",
                "CRA_FAKE_DEVICES": "1", "PYTHONPATH": ROOT + os.pathsep + os.path.join(ROOT, "tests")})
    r = subprocess.run(["python3", os.path.join(ROOT, "tests", "soak_gpu.py"), "--seconds", "50", "--seed", "5", "--radii", "3,8", "--no-big"],
                       capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and " 0 failure(s)" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail


@pytest.fixture(scope="module")
def instance_libraries():
    """the fake seam + the product's real instance tables (tests/hostshim/Makefile, flavour_inst): needs the product's HIP objects, which
    __graft_entry__.build() / `make -C clownresampler_amd/csrc` leave in clownresampler_amd/csrc/build"""
    if not os.path.exists(os.path.join(ROOT, "clownresampler_amd", "csrc", "build", "cr_kernels.o")):
        r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "clownresampler_amd", "csrc")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    targets = [os.path.join(BUILD, n) for n in ("libcr_hostshim_inst.so", "libcr_hostshim_inst_asan.so")]
    r = subprocess.run(["make", "-s", "-j4", "-C", SHIM] + targets, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return targets


def _instances(library, steps, sanitized, extra_env=None, timeout=1500):
    env = dict(os.environ)
    env.update({"CLOWNRESAMPLER_AMD_LIBRARY": library, "CRA_FAKE_DEVICES": "1", "PYTHONPATH": ROOT + os.pathsep + os.path.join(ROOT, "tests")})
    env.pop("CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES", None)
    if sanitized:
        asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
        stdcxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True, check=True).stdout.strip()
        env.update({"LD_PRELOAD": asan + " " + stdcxx, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    env.update(extra_env or {})
    r = subprocess.run(["python3", os.path.join(SHIM, "instances_driver.py")] + steps, capture_output=True, text=True, timeout=timeout, env=env)
    tail = (r.stdout + r.stderr)[-5000:]
    assert r.returncode == 0 and "instances driver: all passed" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    return r


def test_specialised_launch_arithmetic_against_the_oracle(instance_libraries):
    """every plan with the kernel, variant and geometry it has on the GPU (the product's own instance tables answer), every
    launch checked by the product's own launch function, validated range by range and computed by a scalar model of its arguments: the
    device-resident cases, the host-pointer cases (staged and "page-locked"), the segment lists GPUTEST_r05 died in, dual mono, k_seg, k_int,
    brief shapes, padded tiles, the callback forms - all equal to the oracle, the plan cache at two plans."""
    r = _instances(instance_libraries[0], [], False)
    for step in ("cases", "bulk", "direct", "segments", "long", "callback"):
        assert "ok " + step in r.stdout, r.stdout[-3000:]
    assert "(3, 1, 65535)" in r.stdout and "(4, 1, 30)" in r.stdout     # k_up2 and k_wave2 plans were made
    _instances(instance_libraries[0], ["long_up2"], False, {"CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES": "0"})


def test_specialised_launch_arithmetic_under_address_sanitizer(instance_libraries):
    """... and the same under AddressSanitizer + UBSan: "device memory" is malloc of the exact size, so a launch whose input, output, rows image or
    ticket block leaves its allocation by one byte - or names an image that an eviction has freed - is reported; so is every 32-bit
    descriptor field that overflows on the way (UBSan)."""
    _instances(instance_libraries[1], ["cases", "direct", "segments", "long", "callback"], True, timeout=2400)
    _instances(instance_libraries[1], ["long_up2"], True, {"CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES": "0"})


def test_c_driver_with_the_real_instance_tables_under_sanitizers(instance_libraries):
    """The C driver of the plain fake seam (threads: concurrent callers over a small plan cache, device-resident launches from many threads, the
    compute-ahead and download helper threads, failure injection at malloc / launch / copy / sync) against the library with the product's REAL
    instance tables: plain, under ASan + UBSan with leak detection, under ThreadSanitizer."""
    targets = [os.path.join(BUILD, n) for n in ("driver_inst", "driver_inst_asan", "driver_inst_tsan")]
    r = subprocess.run(["make", "-s", "-j4", "-C", SHIM] + targets, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    _run(targets[0], {}, 300)
    r = _run(targets[1], {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}, 900)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
    r = _run(targets[2], {"TSAN_OPTIONS": "halt_on_error=0:exitcode=66"}, 1200)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
