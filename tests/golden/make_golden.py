#!/usr/bin/env python3
"""Regenerates tests/golden/bootstrap.json from the reference's embedded
bootstrap file (pkg/spicedb/bootstrap.yaml:1-40).  Run in the build container
only (/root/reference does not exist on the GPU box); the output is committed.

kats.json is hand-written: each entry restates, as a request/response sequence
at the v1.PermissionsServiceClient seam, an assertion the reference's own tests
make (source file:line in each entry's "source").
"""
import json
import os
import sys

import yaml

REF = "/root/reference/pkg/spicedb/bootstrap.yaml"


def main():
    with open(REF) as f:
        doc = yaml.safe_load(f)
    out = {
        "source": "pkg/spicedb/bootstrap.yaml:1-40",
        "schema": doc["schema"],
        "relationships": [l.strip() for l in doc["relationships"].splitlines() if l.strip()],
    }
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "bootstrap.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote bootstrap.json:", len(out["schema"]), "schema bytes,", len(out["relationships"]), "relationships")


if __name__ == "__main__":
    sys.exit(main())
