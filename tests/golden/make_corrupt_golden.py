#!/usr/bin/env python3
"""Generates tests/golden/corrupt_decode.json: what the REAL reference decoder (oracle/_ref, CRC check off) makes of 60
deterministically damaged copies of one .lnn stream -- its result code and, where it returns OK, the sha256 of the PCM.
The reference has undefined behaviour on some damaged streams (it can crash), so every trial runs in a child process;
crashed trials are recorded as such and skipped by the test.  Run here (needs oracle/_ref); the fixture travels.

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


CHILD = r"""
import sys, hashlib, json
sys.path.insert(0, %r)
import numpy as np
from refs import Reference
data = open(sys.argv[1], 'rb').read()
ret, pcm = Reference().decode_whole(data, check_crc=0)
print(json.dumps({"ret": int(ret), "sha256": hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest()}))
"""


def main():
    from refs import Reference
    x, rng = stream_and_damage()
    good = Reference().encode_whole(x, 16, 44100, 2048, 7, True)
    out = {"good_sha256": hashlib.sha256(good).hexdigest(), "trials": []}
    tmp = os.path.join(HERE, "_corrupt_tmp.lnn")
    for t in range(NTRIALS):
        bad = damaged(good, rng)
        open(tmp, "wb").write(bad)
        r = subprocess.run([sys.executable, "-c", CHILD % os.path.dirname(HERE), tmp], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        if r.returncode != 0:
            out["trials"].append({"ret": None, "crashed": True})
        else:
            out["trials"].append(json.loads(r.stdout.strip().splitlines()[-1]))
    os.remove(tmp)
    json.dump(out, open(os.path.join(HERE, "corrupt_decode.json"), "w"), indent=1)
    ok = sum(1 for t in out["trials"] if t.get("ret") == 0)
    print(f"{ok} of {NTRIALS} damaged streams decode OK in the reference, {sum(1 for t in out['trials'] if t.get('crashed'))} crash it")


if __name__ == "__main__":
    main()
