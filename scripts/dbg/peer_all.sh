#!/bin/bash
for sc in "1024 13" "256 4"; do echo "=== ipc $sc"; timeout -k 5 200 python3 scripts/dbg/peer_ipc.py $sc 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" | tail -16; done
