#!/bin/bash
for v in default d10 d12; do for sl in 2 3 4; do export CGRT_SUB_LEAF=$sl; bash tools/exp_variants.sh $v; done; done
