import sys, time, os, tempfile, numpy as np
sys.path.insert(0, ".")
from gdmix_amd import synthetic
from gdmix_amd.io import native_reader as nr
from gdmix_amd.io.grouped_reader import write_grouped_partition, resolve_input_files
E = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
b = synthetic.make_batch(E, 16, 4, 1024, seed=1)
with tempfile.TemporaryDirectory() as d:
    nfiles = 8; per = (E + nfiles - 1) // nfiles
    for i in range(nfiles):
        write_grouped_partition(os.path.join(d, f"part-{i:05d}.tfrecord"), b.select(np.arange(i * per, min(E, (i + 1) * per))), "ent", "bag", weight_column_name=None)
    files = resolve_input_files(d)
    import gdmix_amd.io.native_reader as m
    orig_split = m._split_ids
    tsplit = [0.0]
    def timed_split(*a):
        t = time.perf_counter(); r = orig_split(*a); tsplit[0] += time.perf_counter() - t; return r
    m._split_ids = timed_split
    for th in (8, 32, 64, 128):
        best = None
        for _ in range(3):
            tsplit[0] = 0.0
            t = time.perf_counter()
            r = nr.read_grouped_files(files, "ent", "bag", "offset", "uid", "response", None, 1024, True, th)
            dt = time.perf_counter() - t
            if best is None or dt < best[0]: best = (dt, tsplit[0])
        print(f"threads {th:3d}: {best[0]*1e3:7.1f} ms  ({E/best[0]/1e6:.2f} M ent/s)  of which id strings {best[1]*1e3:.1f} ms")
