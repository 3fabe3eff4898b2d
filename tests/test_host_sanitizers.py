"""The host side of rollout ingestion under the sanitizers (VERDICT r04 "no sanitizer build of the host side").

mjrl_amd/csrc/host_ingest.h -- persistent gather pools, the converting gather, per-path sums, the asynchronous staging jobs and
their hand-over to the copy queue -- is plain C++ behind three device hooks; tests/c/host_san.cpp binds the hooks to memcpy, drives
every entry point from four caller threads at once (what train_step does: two staging jobs and the trainer's own gathers in flight
together) and checks the results.  Built and run here under -fsanitize=address,undefined and under -fsanitize=thread.
(The ThreadSanitizer build found a real race on HostPool::owner between two first-time callers: fixed in r05.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "host_san.cpp")


@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]),
                                        ("tsan", ["-fsanitize=thread"])])
def test_host_ingestion_under_sanitizers(tmp_path, name, flags):
    cxx = os.environ.get("CXX", "g++")
    if shutil.which(cxx) is None:
        pytest.skip("no C++ compiler")
    exe = str(tmp_path / ("host_" + name))
    b = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-pthread"] + flags + [SRC, "-o", exe], capture_output=True, text=True, cwd=os.path.dirname(SRC))
    if b.returncode != 0 and "sanitize" in b.stderr and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain has no %s runtime" % name)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=600, env=env)
    # (the pools are leaked on purpose -- no joins in static destructors -- hence detect_leaks=0)
    assert r.returncode == 0 and "host_san ok" in r.stdout and "Sanitizer" not in r.stderr, (r.stdout[-500:], r.stderr[-4000:])
