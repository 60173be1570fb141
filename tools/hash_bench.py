"""Host BLAKE2b throughput of the library's four compression functions (0 portable, 1 AVX2, 2 BMI2-compiled scalar, 3 AVX-512) and of
hashlib on this machine's CPU:  python3 tools/hash_bench.py   (the library picks by timing them on first use)."""
import sys,time,hashlib,os
sys.path.insert(0,'tests')
from conftest import load_package
pkg=load_package()
L=pkg.load_library()
data=os.urandom(256<<20)
for impl in (0,1,2,3,0,1,2,3):
    L.ptau_blake2b_implementation(impl)
    t=time.time(); h=pkg.calculate_hash(data); dt=time.time()-t
    print(impl, "(ran as %d)" % L.ptau_blake2b_implementation(-1), "  %.2f GB/s"%(len(data)/dt/1e9))
t=time.time(); hashlib.blake2b(data).digest(); print("hashlib %.2f GB/s"%(len(data)/(time.time()-t)/1e9))
