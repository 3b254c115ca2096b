"""Sanitizers on the HOST code: the threaded BGZF reader, the reference parsers and the oracle run under
AddressSanitizer + UBSan and under ThreadSanitizer (tests/sanitize/), on the CPU only."""
import os
import subprocess

import pytest

from irfinder_amd.synth.bam import bgzf_compress
from util import small_world

HERE = os.path.dirname(os.path.abspath(__file__))
SAN = os.path.join(HERE, "sanitize")


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    ann, sim = small_world(seed=11)
    raw = sim.bam(0, 6000)
    paths = {"raw": d / "s.raw.bam", "z": d / "s.bam"}
    paths["raw"].write_bytes(raw)
    paths["z"].write_bytes(bgzf_compress(raw, block=0x4000))          # many members: the four-wave pool path
    for name, text in zip(("cover", "sj", "rc", "roi"), ann.ref_texts()):
        paths[name] = d / ("ref-%s.txt" % name)
        paths[name].write_text(text)
    paths["out"] = d / "out"
    paths["out"].mkdir()
    return paths


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_code_under_sanitizer(kind, sample):
    subprocess.check_call(["make", "-C", SAN, kind], stdout=subprocess.DEVNULL)
    # (verify_asan_link_order=0: the environment may preload a library of its own in front of the sanitizer runtime)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    exe = os.path.join(SAN, "_build", "host_%s" % kind)
    r = subprocess.run([exe] + [str(sample[k]) for k in ("raw", "z", "cover", "sj", "rc", "roi", "out")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host_sanitize ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr
    assert os.path.exists(os.path.join(str(sample["out"]), "IRFinder-IR-nondir.txt"))
