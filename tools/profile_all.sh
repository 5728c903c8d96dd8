# Re-creates every rocprofv3 record under profiles/ (run on the GPU box: gpurun -- "bash tools/profile_all.sh"; then, here,
# python tools/collect_profiles.py gpurun_out/r4final).  --pmc passes are separate runs with no tracing domain beside them.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r4final && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_cfg2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-kernel-probes > $O/bench_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_cfg3 -- python3 bench.py --config cfg3 --steps 2 --warmup 0 --no-cpu-baseline --no-kernel-probes > $O/bench_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tick -- python3 tools/profile_tick.py --episodes 2 > $O/tick.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tick_f32 -- python3 tools/profile_tick.py --episodes 2 --float-obs > $O/tick_f32.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc1 -- python3 tools/profile_tick.py > $O/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc2 -- python3 tools/profile_tick.py > $O/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc3 -- python3 tools/profile_tick.py > $O/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf1 -- python3 tools/profile_tick.py --float-obs > $O/pmcf1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcf2 -- python3 tools/profile_tick.py --float-obs > $O/pmcf2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gru_t -- python3 tools/profile_gru.py > $O/gru_t.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --output-format csv -d $O/gru_c -- python3 tools/profile_gru.py > $O/gru_c.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/grug_t -- python3 tools/profile_gru.py --grouped > $O/grug_t.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --output-format csv -d $O/grug_c -- python3 tools/profile_gru.py --grouped > $O/grug_c.log 2>&1
python3 tools/bench_envs.py > $O/envs.log 2>&1
grep "graph-replayed" $O/tick.log $O/tick_f32.log
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
ls $O
