cd /root/repo
for flags in "--no-fe --no-cli --no-other-workloads --no-strong" "--no-cpu-baseline --no-fe --no-cli --no-other-workloads --no-strong" "--no-cli --no-other-workloads --no-strong" "--no-other-workloads --no-strong"; do
python bench.py $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['detail']['host_handover']
print('$flags: handover %.1f M ent/s (%.2f ms/partition)  with index %.1f M   step %.2f ms' % (h['entities_per_s']/1e6, h['ms_per_partition'], h['with_feature_index']['entities_per_s']/1e6, d['ms_per_step']))"
done
