B="python bench.py --workload c4 --precision f64 --steps 300 --warmup 30 --no-cpu-baseline"
bash tools/gpu_steps.sh \
 "r5_valu_f64|120|tools/valu_f64" \
 "r5_s1_gputests|900|python -m pytest tests/ -x -q -m gpu" \
 "r5_f64_default|200|$B" \
 "r5_f64_strict|200|MVRL_LIB=variants_build/libmvrl_f64strict.so $B" \
 "r5_f64_fast|200|MVRL_LIB=variants_build/libmvrl_f64fast.so $B" \
 "r5_f64_nopark|200|MVRL_LIB=variants_build/libmvrl_f64nopark.so $B" \
 "r5_audit_c4_exact|400|python tests/audit/episode_audit.py c4 65536 250" \
 "r5_audit_c4_t32|400|MVRL_LIB=variants_build/libmvrl_flowt32.so python tests/audit/episode_audit.py c4 65536 250"
