import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, sparta_amd as sa
from oracle import oracle as O
m = sa.gen.uniform_random(640,640,20000,seed=640+640+64)
g = np.arange(640)//64
v = sa.VBR().fill_from_CSR_inplace(m,g,64)
n=128
B = sa.gen.dense_rhs(v.cols,n,seed=3)
Co = O.vbr_multiply(v.rows,v.cols,64,v.row_part,v.nzcount,v.jab,v.mab,B,n)
d = v.to_device(0)
C0 = np.zeros(v.rows*n,np.float32); d.spmm_host(B,n,C0,accumulate=False)
C1 = np.zeros(v.rows*n,np.float32); d.spmm_host(B,n,C1,accumulate=True)
print(d.info())
print('overwrite err', np.abs(C0-Co).max(), 'acc err', np.abs(C1-Co).max())
e = np.abs(C1-Co).reshape(n, v.rows)
print('bad cols', np.where(e.max(axis=1)>1e-3)[0][:20], 'bad rows', np.where(e.max(axis=0)>1e-3)[0][:40])
r = C1.reshape(n,v.rows)/np.where(np.abs(Co.reshape(n,v.rows))>1e-3, Co.reshape(n,v.rows), 1)
print(r[:4,:8])
