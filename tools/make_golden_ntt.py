#!/usr/bin/env python3
"""Extract the reference's negacyclic-NTT parameter tables and known-answer vectors into tests/golden/.

Source (data only): /root/reference/src/ntt/params_{N}.rs lines 1-13 -- the constants N, LOGN, NINV, ROOTS,
INVROOTS and the test vectors TESTG / TESTGHAT that the reference's own test
(/root/reference/src/vtfhe/crypto/poly.rs:195-208) checks.  Runs only in the authoring container
(/root/reference does not exist on the GPU box); the JSON it writes is what travels.
For N > 64 ROOTS/INVROOTS are stored as sha256 of their little-endian u64 blobs (they are regenerable:
ROOTS[j] = psi^brev(j), psi = 7^((p-1)/2N)); TESTG/TESTGHAT are always stored in full.
"""
import hashlib, json, os, re, struct, sys

REF = "/root/reference/src/ntt"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

def parse(path):
    txt = open(path).read()
    d = {}
    for name in ("N", "LOGN", "NINV"):
        d[name] = int(re.search(r"pub const %s: \w+ = (\d+);" % name, txt).group(1))
    for name in ("ROOTS", "INVROOTS", "TESTG", "TESTGHAT"):
        body = re.search(r"pub const %s: \[u64; \d+\] = \[([^\]]*)\];" % name, txt).group(1)
        d[name] = [int(x) for x in body.replace("\n", " ").split(",") if x.strip()]
        assert len(d[name]) == d["N"], (name, len(d[name]))
    return d

def main():
    os.makedirs(OUT, exist_ok=True)
    for n in (8, 16, 32, 64, 128, 256, 512, 1024, 2048):
        d = parse(os.path.join(REF, "params_%d.rs" % n))
        for name in ("ROOTS", "INVROOTS"):
            d[name + "_sha256"] = hashlib.sha256(struct.pack("<%dQ" % n, *d[name])).hexdigest()
            if n > 64:
                del d[name]
        d["source"] = "reference src/ntt/params_%d.rs:1-13" % n
        with open(os.path.join(OUT, "ntt_params_%d.json" % n), "w") as f:
            json.dump(d, f)
        print(n, d["ROOTS_sha256"][:8], hashlib.sha256(struct.pack("<%dQ" % n, *d["TESTG"])).hexdigest()[:8],
              hashlib.sha256(struct.pack("<%dQ" % n, *d["TESTGHAT"])).hexdigest()[:8])

if __name__ == "__main__":
    main()
