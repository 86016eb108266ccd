#!/bin/bash
# build the library; non-zero exit (and the first errors) when a source does not compile
out=$(python -m espnet_amd.build 2>&1)
if echo "$out" | grep -q "error"; then echo "$out" | grep -E "error" | head -8; exit 1; fi
echo "$out" | tail -1
