import time, threading, zlib, os
print("cpu.max:", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a')
print("affinity:", len(os.sched_getaffinity(0)))
data = os.urandom(1<<20)*8
comp = zlib.compress(bytes((b%41)+33 for b in os.urandom(4<<20)), 1)
def work(n):
    for _ in range(n): zlib.decompress(comp)
for T in (1,4,16,32,64,128):
    th=[threading.Thread(target=work,args=(20,)) for _ in range(T)]
    t=time.time(); [x.start() for x in th]; [x.join() for x in th]; dt=time.time()-t
    print(T, "threads:", round(T*20*4/dt/1024,2), "GB/s inflate aggregate")
