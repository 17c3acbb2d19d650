#!/bin/bash
# Where the waves of the primary kernel wait: TA / TCP (vector L1) / address translation / LDS / VALU counters, at most two raw
# counters of a block per pass (wider sets are refused by the counter hardware and rocprofv3 aborts), each pass its own run
# under its own timeout, never combined with tracing.  Usage on the GPU box: bash tools/pmc_memory_path.sh
R=$GRAFT_REPO_ROOT
one() { name=$1; shift; echo "-- $name: $*"; timeout -k 10 150 bash $R/tools/pmc_pass.sh $name "$@" || echo "$name: pass failed or timed out (see gpurun_out/$name.err)"; }
one mp_ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
one mp_tcp TCP_GATE_EN1_sum TCP_GATE_EN2_sum
one mp_tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
one mp_tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
one mp_tcp4 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
one mp_tcp5 TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
one mp_tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
one mp_tlb2 TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
one mp_sq SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
one mp_sq2 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT
one mp_sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES
