#!/usr/bin/env python3
"""Git blob hashes of the library's sources, so that a counter summary under profiles/ says which code it was collected on.

  python3 tools/source_stamp.py            -> one line: "# source_hashes: {json: file name -> git blob sha1}" over fedfr_amd/csrc/*.{hip,h,inc}
tools/pmc_traffic.sh / pmc_sq.sh / pmc_step_traffic.sh put that line at the top of their summaries; bench.py compares the hashes of the translation
units a replayed counter belongs to with the working tree and marks the field `"stale": true` when they differ.  (The hash is computed from the file
bytes — sha1("blob <len>\\0" + bytes), what `git hash-object` prints — because the GPU box has no .git directory.)"""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fedfr_amd", "csrc")


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    out = {}
    for pat in ("*.hip", "*.h", "*.inc"):
        for f in sorted(glob.glob(os.path.join(CSRC, pat))):
            out[os.path.basename(f)] = blob_hash(f)[:12]
    return out


def parse_stamp(path):
    """the {file: hash} dict a summary carries, or None (summaries older than round 5 have none)"""
    try:
        for line in open(path):
            if line.startswith("# source_hashes:"):
                return json.loads(line.split(":", 1)[1])
    except (OSError, ValueError):
        pass
    return None


if __name__ == "__main__":
    print("# source_hashes: " + json.dumps(source_hashes(), sort_keys=True))
