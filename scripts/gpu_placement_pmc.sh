#!/bin/bash
# TCC (L2 -> memory) write counters of the same 16384^2 fill into a same-class and a different-class H / P pair.
# Picks whatever spelling of the counters this rocprofv3 offers; results in gpurun_out/pmc/placement_summary.json.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -oE "TCC_EA0_WRREQ[A-Za-z0-9_]*" | sort -u > gpurun_out/pmc/tcc_names.log
pick() { for n in "$@"; do if grep -qx "$n" gpurun_out/pmc/tcc_names.log; then echo -n "$n "; return; fi; done; }
C="$(pick TCC_EA0_WRREQ_sum TCC_EA0_WRREQ)$(pick TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_64B)$(pick TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL)$(pick TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_STALL)"
echo "counters: $C"
rm -rf gpurun_out/pmc/placement
timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc/placement -- python3 scripts/placement_pmc.py > gpurun_out/pmc/placement.log 2>&1
echo "placement rc=$?"
grep -E "candidate|PLACEMENT" gpurun_out/pmc/placement.log | cut -c1-300
python3 scripts/placement_pmc_summary.py gpurun_out/pmc/placement.log gpurun_out/pmc/placement > gpurun_out/pmc/placement_summary.json 2>&1
cat gpurun_out/pmc/placement_summary.json
