set -e
T=r03v10
mkdir -p gpurun_out/$T
python bench.py > gpurun_out/$T/bench_cfg2_full.json 2> gpurun_out/$T/bench_cfg2_full.err
echo "bench done"
bash tools/profile.sh $T cfg2 > gpurun_out/$T/profile_cfg2.log 2>&1; echo "cfg2 done"
bash tools/profile.sh $T cfg3 > gpurun_out/$T/profile_cfg3.log 2>&1; echo "cfg3 done"
bash tools/profile.sh $T cfg4 > gpurun_out/$T/profile_cfg4.log 2>&1; echo "cfg4 done"
bash tools/slice_prof.sh gpurun_out/$T/slice cfg2 > gpurun_out/$T/slice_prof.txt 2>&1; echo "slice prof done"
bash tools/slice_pmc.sh gpurun_out/$T/slice_pmc cfg2 > gpurun_out/$T/slice_pmc.txt 2>&1; echo "slice pmc done"
python tools/stage_alone.py cfg2 > gpurun_out/$T/alone_cfg2.log 2>&1; python tools/stage_alone.py cfg4 > gpurun_out/$T/alone_cfg4.log 2>&1; echo "alone done"
export MGACBAM_LIB=$PWD/build/variants/libmgacbam_trace.so
python tools/trace_gate.py fwd > gpurun_out/$T/trace_k_gate.txt 2>&1
python tools/trace_gate.py bwd > gpurun_out/$T/trace_k_bwd_apply.txt 2>&1
python tools/trace_gate.py pool > gpurun_out/$T/trace_k_pool.txt 2>&1
python tools/trace_r12.py cfg2 > gpurun_out/$T/trace_k_bwd_r12.txt 2>&1
python tools/trace_r12.py cfg3 > gpurun_out/$T/trace_k_bwd_r12_cfg3.txt 2>&1
python tools/trace_head.py fwd > gpurun_out/$T/trace_head_fwd.txt 2>&1
python tools/trace_head.py gx > gpurun_out/$T/trace_head_gx.txt 2>&1
echo "traces done"
