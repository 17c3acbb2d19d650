#!/bin/bash
echo "mode0"; CGRT_PRIMARY_MODE=0 bash tools/exp_variants.sh default
export CGRT_PRIMARY_MODE=1
bash tools/exp_variants.sh default idle32 idle64 default
