# numpy emulation of the wave64 2048-pt real FFT decomposition used by the HIP kernel
import numpy as np
rng = np.random.default_rng(0)
x = rng.standard_normal(2048)
z = x[0::2] + 1j*x[1::2]
W = lambda N, e: np.exp(-2j*np.pi*e/N)
lane = np.arange(64)
# stage A: lane n2 holds z[64*n1+n2]
v = np.stack([z[64*n1 + lane] for n1 in range(16)], axis=1)      # [lane, n1]
A = np.fft.fft(v, axis=1)                                          # [lane n2, k1]
A = A * W(1024, lane[:,None]*np.arange(16)[None,:])
# transpose 1: LDS[k1][n2]
lds = A.T.copy()                                                   # [k1, n2]
k1 = lane >> 2; m2 = lane & 3
b = np.stack([lds[k1, 4*m1 + m2] for m1 in range(16)], axis=1)   # [lane, m1]
B = np.fft.fft(b, axis=1)                                          # [lane, j1]
B = B * W(64, m2[:,None]*np.arange(16)[None,:])
# cross-lane radix-4 over quads
Bq = B.reshape(16, 4, 16)     # [k1, q, j1]
x0,x1,x2,x3 = Bq[:,0],Bq[:,1],Bq[:,2],Bq[:,3]
u0,u1,u2,u3 = x0+x2, x1+x3, x0-x2, x1-x3
u3 = -1j*u3
vq = np.stack([u0+u1, u0-u1, u2+u3, u2-u3], axis=1)  # lanes q=0..3 -> j2 = 0,2,1,3
j2map = np.array([0,2,1,3])
Z = np.zeros(1024, complex)
for L in range(64):
    for j1 in range(16):
        k = (L>>2) + 16*j1 + 256*j2map[L&3]
        Z[k] = vq.reshape(64,16)[L, j1]
print("complex fft err", np.abs(Z - np.fft.fft(z)).max())
# unpack
X = np.zeros(1025, complex)
Wk = W(2048, np.arange(1025))
for k in range(0, 512):
    kk = (1024 - k) & 1023
    E = 0.5*(Z[k] + np.conj(Z[kk])); O = -0.5j*(Z[k] - np.conj(Z[kk])); T = Wk[k]*O
    X[k] = E + T; X[1024-k] = np.conj(E - T)
X[512] = np.conj(Z[512])
print("rfft err", np.abs(X - np.fft.rfft(x)).max())
