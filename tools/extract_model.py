#!/usr/bin/env python3
"""Extract the trained GNN-VC weights (data, not code) from the reference driver.

The reference ships its trained model as a string literal in its own text
format (`src/GNN_VC.cpp:23`, parsed by `gnn::operator>>`,
`src/gnn_inference.cpp:120-139`).  This tool un-escapes that literal into
`gnn-mwvc_amd/data/mwvc_model.txt` so the engine can load the same weights
through its own parser.  The weights are MIT-licensed data of
KennethLangedal/GNN-MWVC (attribution in data/README.md).

Only runs where /root/reference is mounted (the build container); the output
file is committed so nothing reads the reference at run time.
"""
import hashlib
import pathlib
import re
import sys

REF = pathlib.Path("/root/reference/src/GNN_VC.cpp")
OUT = pathlib.Path(__file__).resolve().parent.parent / "gnn-mwvc_amd" / "data" / "mwvc_model.txt"
EXPECT_MD5 = "ffc28149d57cf3a6f65ecee97230e171"


def main() -> int:
    line = REF.read_text().split("\n")[22]
    m = re.match(r'const string model_data = "(.*)";\s*$', line)
    if not m:
        print("model literal not found at GNN_VC.cpp:23", file=sys.stderr)
        return 1
    text = m.group(1).replace("\\n", "\n")
    if "\\" in text:
        print("unexpected escape in model literal", file=sys.stderr)
        return 1
    md5 = hashlib.md5(text.encode()).hexdigest()
    if md5 != EXPECT_MD5:
        print(f"model text md5 {md5} != {EXPECT_MD5}", file=sys.stderr)
        return 1
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(text)
    print(f"wrote {OUT} ({len(text)} bytes, md5 {md5})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
