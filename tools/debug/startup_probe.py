import time, sys
sys.path.insert(0, "/root/repo")
from portcullis_amd import ffi
t0 = time.perf_counter(); n = ffi.device_count(); t1 = time.perf_counter()
c = ffi.Context(0, "FR"); t2 = time.perf_counter()
c2 = ffi.Context(0, "FR"); t3 = time.perf_counter()
c.set_refs([1000]); c.upload_contig(0, b"A" * 1000); t4 = time.perf_counter()
print(f"device_count (HIP init) {t1 - t0:.3f} s, first pjb_create {t2 - t1:.3f} s, second {t3 - t2:.3f} s, first upload (first kernels) {t4 - t3:.3f} s")
c.close(); c2.close()
