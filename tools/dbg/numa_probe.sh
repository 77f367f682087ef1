cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
cat /sys/devices/system/node/node*/cpulist
which numactl
for node in 0 1; do
  cpus=$(cat /sys/devices/system/node/node$node/cpulist)
  echo "== taskset node $node ($cpus)"
  taskset -c $cpus python tools/dbg/pcie_probe.py 2>&1 | grep -v amdgpu.ids
done
