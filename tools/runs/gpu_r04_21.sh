#!/bin/bash
# round 4, call 21: the repeated-name memo of the string path, on and off, on the three batch shapes (same box, alternating)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do for K in 1 0; do
  echo "ACL_INTERN_REPEAT=$K"; ACL_INTERN_REPEAT=$K timeout 600 python tools/string_shapes.py 2>&1 | tail -1 | cut -c1-600
done; done
