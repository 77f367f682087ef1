#!/bin/bash
# gpurun_out/r03ev (tools/evidence_session.sh) -> the round's files under profiles/ (run here, after the GPU session)
E=gpurun_out/r03ev; R=${1:-r03}
cp $E/c2_summary.txt profiles/${R}_final_c2_1m.txt
cp $E/latest_traffic.json profiles/latest_traffic.json
{ echo "# Zipf 1 M entities (bench.py --workload zipf --entities 1000000): team kernels, kernel-trace duration and FETCH_SIZE / WRITE_SIZE (separate --pmc passes) per launch; tools/zipf_pmc.sh"
  echo "# order of the launches: one-workgroup class (8 765 entities), 128 teams (789), 32 teams (74), 8 teams (6); warm-up step first, measured step second"
  echo "# round 2 (profiles/r02_zipf_1m.txt): 68.4 ms / 133.8 GB fetched for the one-workgroup class, 55.7 / 110.2, 49.1 / 58.9, 28.5 / 21.5"
  cat $E/zipf_pmc.txt
  echo; echo "# the same workload, bench line (3 steps after 1 warm-up): per-class times from HIP events"
  python tools/bench_summary.py $E/zipf_bench.json | sed "s#== $E/zipf_bench.json:#== zipf (plain run):#" | cut -c1-220; } > profiles/${R}_zipf_1m.txt
{ echo "# bench.py --workload <w> --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads under rocprofv3 --kernel-trace (tools/profile_workload.sh), one MI355X"
  echo "# (pack + solve per step on a resident batch; per-class solve times from HIP events; times under the profiler are a few per cent above a plain run)"
  python tools/bench_summary.py $E/ml20m_user_bench_line.json $E/ml20m_movie_bench_line.json $E/c5share_bench_line.json $E/zipf_bench.json | sed "s#== $E/\([a-z0-9_]*\)_bench_line.json:#== \1:#; s#== $E/zipf_bench.json:#== zipf (plain run):#" | cut -c1-230
  echo; echo "# C2 headline of the same session: profiles/${R}_final_c2_1m.txt"; } > profiles/${R}_bench_workloads.txt
for w in ml20m_user ml20m_movie c5share; do { echo "# tools/profile_workload.sh $w $w pmc: rocprofv3 --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE / SQ counters in separate --pmc passes"; grep -v "^$" $E/${w}_summary.txt | awk '/^# launch geometry/{skip=1} /^# PMC pass/{skip=0} !skip' | grep -E "^#|^kernel|gdmix::|rocprim" | cut -c1-170 | head -150; } > profiles/${R}_prof_$w.txt; done
{ echo "# host path (GPU box's host cores): tools/io_parts.py 1000000 (reader scaling), tools/e2e_bench.py 1000000 8 (CLI in process, phases)"; grep -v "^INFO" $E/host_path.txt | grep -v amdgpu.ids | cut -c1-300; } > profiles/${R}_host_path.txt
