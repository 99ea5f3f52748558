#!/bin/bash
# round 4, call 16: long differential-fuzz campaigns after the round's changes -- the C4 schema and the combine schema (incl. `.all()`), with id
# recycling at work (quarantine 0, a stream of never-seen names), background compactions, write bursts, and the expiring-keys campaign
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 600 2>&1 | tail -4 | cut -c1-400
run() { echo "== $*"; timeout 900 python tools/fuzz_gpu.py "$@" 2>&1 | tail -1 | cut -c1-700; }
run --seed 51 --steps 600 --recycle
run --seed 52 --steps 500 --recycle --compact-early
run --seed 53 --steps 500 --recycle --schema combine
run --seed 54 --steps 400 --recycle --schema combine --compact-early
run --seed 55 --steps 150 --recycle --burst 300 --universe 3
run --seed 56 --steps 600
run --seed 57 --steps 600 --schema combine
run --seed 58 --steps 800 --expiry
