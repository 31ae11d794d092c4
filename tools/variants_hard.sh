#!/bin/bash
for so in km_amd/variants/*.so; do echo "== $(basename $so .so)"; KM_LIBRARY=$PWD/$so timeout -k 10 300 python3 tools/hard_only.py 2>&1 | grep -v amdgpu.ids | tail -2; done
