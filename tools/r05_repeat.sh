#!/bin/bash
# repeated timing (run-to-run spread) of the libraries named on the 64-chain C3 batch: N runs each, interleaved
cd "$GRAFT_REPO_ROOT"; N=${N:-5}
for i in $(seq $N); do for n in "$@"; do echo "$n $(DA4ML_HIP_LIB=ab_libs/lib_$n.so timeout 90 python tests/gpu_profile.py 256 ${B:-64} | head -1 | sed 's/.*us\/iter //')"; done; done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; c[$1]++} END{for(k in a) printf "%-10s mean %.2f  runs%s\n", k, s[k]/c[k], a[k]}'
