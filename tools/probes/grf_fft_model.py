import numpy as np
rs = np.random.RandomState(0)

def fft5(x):  # x: list of 5 complex -> forward DFT
    c1, c2 = np.cos(2*np.pi/5), np.cos(4*np.pi/5)
    s1, s2 = np.sin(2*np.pi/5), np.sin(4*np.pi/5)
    t1, t2, t3, t4 = x[1]+x[4], x[2]+x[3], x[1]-x[4], x[2]-x[3]
    X0 = x[0]+t1+t2
    m1 = x[0] + c1*t1 + c2*t2
    m2 = x[0] + c2*t1 + c1*t2
    u1 = s1*t3 + s2*t4
    u2 = s2*t3 - s1*t4
    # X1 = m1 - i u1 ; X4 = m1 + i u1 ; X2 = m2 - i u2 ; X3 = m2 + i u2
    return [X0, m1 - 1j*u1, m2 - 1j*u2, m2 + 1j*u2, m1 + 1j*u1]

def fft10(x):
    e = fft5([x[0], x[2], x[4], x[6], x[8]])
    o = fft5([x[1], x[3], x[5], x[7], x[9]])
    out = [0]*10
    for k in range(5):
        w = np.exp(-2j*np.pi*k/10)
        t = w*o[k]
        out[k] = e[k] + t
        out[k+5] = e[k] - t
    return out

for f, N in ((fft5, 5), (fft10, 10)):
    x = rs.normal(size=N) + 1j*rs.normal(size=N)
    assert np.allclose(f(list(x)), np.fft.fft(x)), N

def fft_n(z, N1, N2):
    """length N1*N2 FFT in place on array z (complex), two-step: index j = c + N2? ... view j = c + N1c?"""
    n = N1*N2
    # step A: for each c in 0..N2-1: FFT-N1 over elements c + N2*r (r = 0..N1-1); result index k (0..N1-1) stored at c + N2*k, times w_n^{c k}
    fa = fft5 if N1 == 5 else fft10
    fb = fft5 if N2 == 5 else fft10
    z = z.copy()
    for c in range(N2):
        y = fa([z[c + N2*r] for r in range(N1)])
        for k in range(N1):
            z[c + N2*k] = y[k]*np.exp(-2j*np.pi*c*k/n)
    # step B: for each k: FFT-N2 over contiguous row (N2*k + c), output q -> X[k + N1*q]
    out = np.empty(n, dtype=complex)
    for k in range(N1):
        y = fb([z[N2*k + c] for c in range(N2)])
        for q in range(N2):
            out[k + N1*q] = y[q]
    return out

for N1, N2 in ((10, 10), (5, 10)):
    n = N1*N2
    x = rs.normal(size=n) + 1j*rs.normal(size=n)
    assert np.allclose(fft_n(x, N1, N2), np.fft.fft(x)), (N1, N2)

def dht_pairs(X, N1, N2, axis):
    """1-D Hartley transform (unnormalised, cas kernel) of every vector of real array X along `axis`, vectors packed two by two."""
    n = X.shape[0]
    Y = X.copy() if axis == 1 else X.T.copy()
    for f in range(n // 2):
        z = Y[2*f] + 1j*Y[2*f+1]
        Z = fft_n(z, N1, N2)
        a, b = Z.real, Z.imag
        for j in range(n // 2 + 1):
            jj = (n - j) % n
            c, d = a[jj], b[jj]
            # DHT_x[j] = (a + c - b + d)/2 ; DHT_y[j] = (b + d + a - c)/2 (factor 1/2 dropped)
            xj, yj = a[j] + c - b[j] + d, b[j] + d + a[j] - c
            xjj, yjj = c + a[j] - d + b[j], d + b[j] + c - a[j]
            Y[2*f][j], Y[2*f+1][j] = xj, yj
            Y[2*f][jj], Y[2*f+1][jj] = xjj, yjj
    return Y if axis == 1 else Y.T

def grf_ref(w, cr=5.0):
    n = w.shape[0]
    k = np.concatenate([np.arange(0, n//2 + 1), np.arange(-(n//2 - 1), 0)])
    kk = np.sqrt(k[:, None]**2 + k[None, :]**2)
    amp = np.zeros((n, n)); amp[kk > 0] = np.sqrt(kk[kk > 0]**(-cr))
    f = np.fft.ifft2(np.fft.fft2(w)*amp).real
    return (f - f.min())/(f.max() - f.min()), amp

for n, (N1, N2) in ((100, (10, 10)), (50, (5, 10))):
    w = rs.normal(size=(n, n))
    ref, amp = grf_ref(w)
    # check 1-D DHT
    H = np.cos(2*np.pi*np.outer(np.arange(n), np.arange(n))/n) + np.sin(2*np.pi*np.outer(np.arange(n), np.arange(n))/n)
    assert np.allclose(dht_pairs(w, N1, N2, 1), 2*(w @ H)), n
    assert np.allclose(dht_pairs(w, N1, N2, 0), 2*(H @ w)), n
    T = dht_pairs(dht_pairs(w, N1, N2, 1), N1, N2, 0)      # H w H (x4)
    T = T*amp
    F = dht_pairs(dht_pairs(T, N1, N2, 1), N1, N2, 0)
    F = (F - F.min())/(F.max() - F.min())
    print(n, "max |field - ref|", np.abs(F - ref).max())
