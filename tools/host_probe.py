import numpy as np, time, os, sys
rs=np.random.RandomState(0); n=100*160000
t=time.time(); a=rs.permutation(n); print('perm', time.time()-t, flush=True)
buf=np.random.rand(4096,9).astype(np.float32)
for d in ('/tmp/wp', '/dev/shm/wp'):
    os.makedirs(d, exist_ok=True)
    t=time.time()
    for k in range(1000): np.save(f'{d}/data_{k}.npy', buf)
    print(d, '1000 x np.save 147KB', time.time()-t, flush=True)
    import shutil; shutil.rmtree(d)
print(os.cpu_count(), len(os.sched_getaffinity(0)))
