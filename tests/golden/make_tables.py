#!/usr/bin/env python3
"""Extract the three live lookup tables of the reference as plain-text data.

Source of the numbers: the `static` array literals in the reference's
src/denoising/mod.rs (lines 55, 59, 61).  Only the numeric tokens are taken
(data, not code); every token is written verbatim, one per line, so that the
decimal -> f32 rounding is done by the consumer (strtof), exactly like rustc
rounds the literal.

The script also checks that the two tables that have a twin under the
reference's data/ directory are token-identical to it, and prints sha256 sums
that tests/golden/tables.sha256 pins.

Run from the repo root, in the build container only (needs /root/reference):
    python tests/golden/make_tables.py
"""
import hashlib
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "raytracing_amd", "data")
NUM = re.compile(r"-?\d+\.\d+(?:[eE][-+]?\d+)?|-?\d+(?:[eE][-+]?\d+)")


def tokens_after_equals(line):
    return NUM.findall(line.split("= &[", 1)[1])


def main():
    with open(os.path.join(REF, "src/denoising/mod.rs")) as f:
        lines = f.read().split("\n")
    want = {
        "disc_blue_noise_64.txt": ("disc_blue_noise", 128),
        "spherical_blue_noise_256.txt": ("spherical_blue_noise_256", 768),
        "screenspace_blue_noise_128x128.txt": ("screenspace_blue_noise_128_128", 16384),
    }
    twins = {
        "spherical_blue_noise_256.txt": "data/random_in_unit_sphere_256.txt",
        "screenspace_blue_noise_128x128.txt": "data/blue_noise_screenspace_64_64.txt",
    }
    os.makedirs(OUT, exist_ok=True)
    sums = []
    for fname, (ident, count) in want.items():
        line = [l for l in lines if ("static " + ident + ":") in l]
        assert len(line) == 1, ident
        toks = tokens_after_equals(line[0])
        assert len(toks) == count, (ident, len(toks))
        if fname in twins:
            with open(os.path.join(REF, twins[fname])) as f:
                twin = NUM.findall(f.read())
            assert twin == toks, "twin mismatch for " + fname
        body = ("\n".join(toks) + "\n").encode()
        with open(os.path.join(OUT, fname), "wb") as f:
            f.write(body)
        sums.append("%s  %s" % (hashlib.sha256(body).hexdigest(), fname))
    text = "\n".join(sums) + "\n"
    with open(os.path.join(os.path.dirname(__file__), "tables.sha256"), "w") as f:
        f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
