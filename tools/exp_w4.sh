#!/bin/bash
for sl in 2 3 4; do export CGRT_SUB_LEAF=$sl; bash tools/exp_variants.sh default w2; done
