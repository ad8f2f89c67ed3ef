#!/usr/bin/env python3
"""Generates tests/golden/corrupt_decode.json: what the REAL reference decoder (oracle/_ref, CRC check off) makes of 60
deterministically damaged copies of one .lnn stream -- its result code and, where it returns OK, the sha256 of the PCM.
The reference has undefined behaviour on some damaged streams (reads past the end of the data, shifts by 32 or more, wild
partition orders), so every trial runs in a child process: oracle/_ref/ref_decode_san, the reference decoder built with
AddressSanitizer + UBSan.  A sanitizer report marks the trial "undefined" (skipped by the test); the others carry the result
code and an FNV-1a-64 of the decoded planes.  Run here (needs oracle/_ref); the fixture travels.

    python tests/golden/make_corrupt_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

NTRIALS, SEED = 60, 777


def stream_and_damage():
    """the good stream's input and the list of (position, bit) flips per trial -- shared with the test"""
    import numpy as np
    from signals import music
    rng = np.random.default_rng(SEED)
    x = music(2, 6 * 2048 + 300, 16, seed=9)
    return x, rng


def damaged(good, rng):
    bad = bytearray(good)
    for _ in range(int(rng.integers(1, 4))):
        pos = int(rng.integers(41, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
    return bytes(bad)


def fnv_planes(pcm):
    """FNV-1a-64 over the int32 planes, little-endian bytes, channel after channel (what oracle/ref_decode_san.c prints)"""
    import numpy as np
    h = 0xcbf29ce484222325
    for byte in np.ascontiguousarray(pcm, dtype="<i4").tobytes():
        h = ((h ^ byte) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def main():
    from refs import Reference
    san = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_decode_san")
    x, rng = stream_and_damage()
    good = Reference().encode_whole(x, 16, 44100, 2048, 7, True)
    out = {"good_sha256": hashlib.sha256(good).hexdigest(), "trials": []}
    tmp = os.path.join(HERE, "_corrupt_tmp.lnn")
    for t in range(NTRIALS):
        bad = damaged(good, rng)
        open(tmp, "wb").write(bad)
        # the reference decoder under ASan + UBSan (shift exponents, bounds): a sanitizer report = the result is undefined
        r = subprocess.run([san, tmp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        if r.returncode != 0 or not r.stdout.startswith("ret "):
            out["trials"].append({"defined": False, "why": ([l for l in r.stderr.strip().splitlines() if "ERROR" in l or "runtime error" in l] or ["crash"])[0][:200]})
            continue
        parts = r.stdout.split()
        out["trials"].append({"defined": True, "ret": int(parts[1]), "fnv": parts[3]})
    os.remove(tmp)
    json.dump(out, open(os.path.join(HERE, "corrupt_decode.json"), "w"), indent=1)
    ok = sum(1 for t in out["trials"] if t.get("ret") == 0)
    print(f"{ok} of {NTRIALS} damaged streams decode OK with defined behaviour, {sum(1 for t in out['trials'] if not t['defined'])} are undefined in the reference")


if __name__ == "__main__":
    main()
