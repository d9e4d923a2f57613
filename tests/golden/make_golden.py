#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REAL reference, in the build container only.

What it runs is /root/reference compiled in place by oracle/Makefile (oracle/_ref/): the reference
header behind oracle/ref/ref_wrap.c, the reference's own harnesses tests/test-{low,high}-level.c and a
dr_flac-based decoder.  Nothing from /root/reference is copied except DATA: the decoded PCM of its
test fixture tests/test.flac and its golden output tests/test3 (SURVEY.md section 4 explains why that
file is stale and what it still pins).

Outputs (all committed):
  test_flac_s16le_2ch_192000.pcm   tests/test.flac decoded to interleaved s16-LE by dr_flac
  ref_test3.bin                    copy of the reference's tests/test3 (== tests/test4), int32-LE
  golden.json                      per-case known answers (tests/_cases.py CASES), config scalars,
                                   table hashes, harness sha256s, short full-output vectors
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import _checkers as ck  # noqa: E402
import _cases  # noqa: E402

REF = ck.REFERENCE_DIR
REFBIN = os.path.join(ck.ORACLE_DIR, "_ref")


def sha_file(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def main():
    assert os.path.isdir(REF), "golden fixtures can only be regenerated where /root/reference exists"
    ck.build_checkers()
    gold = {"generator": "tests/golden/make_golden.py", "source": "real reference compiled in place (oracle/_ref)"}

    # 1. the reference's input fixture, decoded the way its harness decodes it
    pcm_path = _cases.FLAC_PCM
    info = subprocess.run([os.path.join(REFBIN, "flac2pcm"), os.path.join(REF, "tests", "test.flac"), pcm_path],
                          check=True, capture_output=True, text=True).stdout.split()
    gold["flac"] = {"channels": int(info[0]), "file_rate": int(info[1]), "frames": int(info[2]),
                    "flac_sha256": sha_file(os.path.join(REF, "tests", "test.flac")), "pcm_sha256": sha_file(pcm_path)}

    # 2. the reference's golden output file (data)
    shutil.copyfile(os.path.join(REF, "tests", "test3"), os.path.join(HERE, "ref_test3.bin"))
    gold["ref_test3"] = {"sha256": sha_file(os.path.join(HERE, "ref_test3.bin")),
                         "test4_identical": sha_file(os.path.join(REF, "tests", "test4")) == sha_file(os.path.join(REF, "tests", "test3"))}

    # 3. the reference's own harness binaries on its ctest triples (tests/CMakeLists.txt:25-46) + cfg 1
    harness = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, args in [("cfg1", (44100, 48000, 44100)), ("ctest1", (8000, 44100, 44100)), ("ctest2", (8000, 44100, 8000)),
                           ("ctest3", (44100, 8000, 44100)), ("ctest4", (44100, 8000, 8000))]:
            for level in ("low", "high"):
                outp = os.path.join(tmp, "o.bin")
                subprocess.run([os.path.join(REFBIN, "test-%s-level" % level), os.path.join(REF, "tests", "test.flac"), outp] +
                               [str(a) for a in args], check=True, capture_output=True)
                harness["%s_%s" % (name, level)] = {"sha256": sha_file(outp), "bytes": os.path.getsize(outp)}
    gold["harness"] = harness

    # 4. tables
    gold["table"] = {}
    for r in _cases.RADII:
        t = ck.reference(r).table()
        gold["table"][str(r)] = {"len": int(len(t)), "sha256_i32le": hashlib.sha256(t.astype("<i4").tobytes()).hexdigest(),
                                 "sum": int(t.sum()), "abs_sum": int(np.abs(t).sum()), "t512": int(t[512]),
                                 "min": int(t.min()), "max": int(t.max())}

    # 5. ratio / configuration scalars
    gold["config"] = {}
    for r in _cases.RADII:
        ref = ck.reference(r)
        rows = []
        for (i, o, l) in _cases.CONFIG_TRIPLES:
            st = ck.LowLevel()
            for f in ("pos_int", "pos_frac", "increment"):
                setattr(st, f, 0x5A5A5A5A)
            st.cfg.stretched_radius = st.cfg.radius_frames = st.cfg.radius_delta = st.cfg.table_step = 0x5A5A5A5A
            ok, st = ref.low_init(2, i, o, l, st)
            rows.append({"rates": [i, o, l], "ok": int(ok), "state": [int(v) for v in st.astuple()],
                         "ratio_in_out": int(ref.ratio(i, o))})
        gold["config"][str(r)] = rows

    # 6. per-case known answers (+ full outputs of the short cases)
    gold["cases"] = {}
    vectors = {}
    for case in _cases.CASES:
        ref = ck.reference(case["radius"])
        res = _cases.run_case(ref, case, keep_output=True)
        out = res.pop("_out")
        gold["cases"][case["name"]] = res
        if out.size <= 1200:
            vectors[case["name"]] = [int(v) for v in out]
        print("%-24s frames=%-9d fnv=%s" % (case["name"], res["frames"], res["fnv"]))
    gold["vectors"] = vectors

    # 7. single-frame known answers incl. a non-zero incoming accumulator (clownresampler.h:1020,1033 "+=" semantics)
    frames = []
    rng = np.random.RandomState(1234)
    for r, rates, ch in [(3, (44100, 48000, 44100), 2), (3, (48000, 44100, 44100), 8), (3, (44100, 8000, 8000), 1), (8, (8000, 96000, 8000), 2)]:
        ref = ck.reference(r)
        ok, cfg = ref.configure(*rates)
        R = int(cfg.radius_frames)
        pcm = ck.noise_pcm((2 * R + 40) * ch, 777 + r)
        for _ in range(6):
            pi = int(rng.randint(0, 40))
            pf = int(rng.choice([0, 1, 65535, int(rng.randint(0, 65536))]))
            acc0 = [int(v) for v in rng.randint(-100000, 100000, size=ch)] if _ % 2 else [0] * ch
            out = ref.frame(cfg, ch, pcm, pi, pf, acc0)
            frames.append({"radius": r, "rates": list(rates), "channels": ch, "seed": 777 + r, "pcm_frames": 2 * R + 40,
                           "pos_int": pi, "pos_frac": pf, "acc_in": acc0, "acc_out": [int(v) for v in out]})
    gold["single_frames"] = frames

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
