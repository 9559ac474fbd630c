#!/bin/bash
# lab: round-robin dealing of the hub units against consecutive units, on the hub parts
for cfg in 5 1 c4; do python scripts/lab/r4_hub_parts.py $cfg,only SPARTA_HUB_DEAL=0 2>&1 | grep -v "Warning\|amdgpu.ids\|sparse rows:" | cut -c1-420; done
