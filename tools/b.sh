#!/bin/bash
# build the library; non-zero exit (and the first errors) when a source does not compile
cd "$(dirname "$0")/.." || exit 1
out=$(python -m espnet_amd.build 2>&1)
if echo "$out" | grep -qi "error"; then echo "$out" | grep -E "error" | head -8; exit 1; fi
echo "$out" | tail -1
