"""AddressSanitizer + UBSan over the CPU oracle and the product's pure-host sources
(voice algebra, text front half, RIFF writer; the launch policy: voice analysis, time-split grids,
block planner).  GPU ASan is not available on the pool."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_oracle_and_host_sources_under_asan_ubsan(tmp_path):
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1",
           "-ffp-contract=off"]
    csrc = os.path.join(ROOT, "grail-rs_amd", "csrc")
    objs = []
    for name, cc, std in (("voice_host.cpp", "g++", "-std=c++17"), ("text_front.cpp", "g++", "-std=c++17")):
        o = str(tmp_path / (name + ".o"))
        subprocess.check_call([cc, std, *san, "-c", os.path.join(csrc, name), "-o", o])
        objs.append(o)
    for name in (os.path.join(ROOT, "oracle", "grail_oracle.c"), os.path.join(ROOT, "tests", "sanitize_driver.c")):
        o = str(tmp_path / (os.path.basename(name) + ".o"))
        subprocess.check_call(["gcc", "-std=c11", *san, "-c", name, "-o", o])
        objs.append(o)
    exe = str(tmp_path / "sanitize_driver")
    subprocess.check_call(["g++", *san, *objs, "-o", exe, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitize driver: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.timeout(300)
def test_launch_policy_under_asan_ubsan(tmp_path):
    """launch_plan.cpp + voice_analysis.cpp make no HIP call: built with g++ and the sanitizers, then fed random and
    hostile arguments through grail_plan_blocks / grail_time_split_grid / grail_fast_sharpness / grail_time_split_warmup
    (tests/sanitize_plan_driver.cpp checks the invariants of what comes back)."""
    san = ["-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=undefined", "-g", "-O1",
           "-ffp-contract=off", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    csrc = os.path.join(ROOT, "grail-rs_amd", "csrc")
    objs = []
    for name in (os.path.join(csrc, "launch_plan.cpp"), os.path.join(csrc, "voice_analysis.cpp"),
                 os.path.join(csrc, "voice_host.cpp"), os.path.join(ROOT, "tests", "sanitize_plan_driver.cpp")):
        o = str(tmp_path / (os.path.basename(name) + ".o"))
        subprocess.check_call(["g++", *san, "-c", name, "-o", o])
        objs.append(o)
    exe = str(tmp_path / "sanitize_plan_driver")
    subprocess.check_call(["g++", "-fsanitize=address,undefined", *objs, "-o", exe, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=250)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitize plan driver: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
