#!/usr/bin/env python
"""Can swish be evaluated WITHOUT transcendentals in packed fp16 (VERDICT r4 next #1a)?  swish(y) = h + h tanh(h), h = y / 2; tanh(h) / h
is even, so: a (Lawson-reweighted least-squares, near-minimax) polynomial P in u = h^2 on |h| <= c, the argument clamped to [-c, c] so
that the form saturates to max(y, 0) outside, evaluated the way v_pk_fma_f16 would (every fused multiply-add rounded once to fp16).
Prints, per clamp c and degree d of P: the fit's error in exact arithmetic, and max / rms absolute error of the fp16 evaluation against
float64 swish over y in [-24, 24].  Result: 6e-3 at best in exact arithmetic for d = 4, and the fp16 Horner chain is
unusable (coefficients underflow, terms cancel) — bf16 rounding of the output itself is ~3e-4 rms on swish(N(0,1)) values.   CPU only."""
import numpy as np
from numpy.polynomial import chebyshev as C
def swish(y): return y/(1+np.exp(-y))
def fit_tanh_over_h(c, d, iters=60):
    # weighted least squares on dense Chebyshev nodes in u=h^2, iterate reweighting (Lawson) for near-minimax of error in s = h + h*hc*P(u)
    n=4000
    h=np.cos(np.linspace(0,np.pi,n))*0.5*c+0.5*c   # [0,c]
    h=np.sort(h); h=h[h>1e-6]
    u=h*h
    target=np.tanh(h)/h
    # error in s: h * h * (P - target) -> weight h^2
    w=np.ones_like(h)
    A=np.vander(u,d+1,increasing=True)
    for it in range(iters):
        W=(w*h*h)[:,None]
        coef,_,_,_=np.linalg.lstsq(A*W,target*W[:,0],rcond=None)
        err=np.abs(h*h*(A@coef-target))
        w=w*(0.5+err/err.max()); w/=w.mean()
    return coef, err.max()
def f16(x): return np.asarray(x,dtype=np.float16)
def fma16(a,b,c): return f16(a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64))
def eval16(coef,c,y):
    # y: float32 array of pre-activation; h = y/2 in fp16 (from mix fma: single rounding of exact fp32 h)
    h=f16(np.float32(y)*np.float32(0.5))
    hc=np.minimum(np.maximum(h,f16(-c)),f16(c))
    u=f16(hc.astype(np.float64)*hc.astype(np.float64))
    k=[f16(v) for v in coef]
    p=np.full_like(u,k[-1])
    for v in k[-2::-1]:
        p=fma16(p,u,np.full_like(u,v))
    t=f16(hc.astype(np.float64)*p.astype(np.float64))
    s=fma16(h,t,h)
    return s
if __name__=="__main__":
    y=np.linspace(-24,24,200001).astype(np.float32)
    ref=swish(y.astype(np.float64))
    for c in (3.0,3.5,4.0,4.5,5.0):
        for d in (3,4,5,6):
            coef,e=fit_tanh_over_h(c,d)
            s=eval16(coef,c,y).astype(np.float64)
            err=np.abs(s-ref)
            # error vs fp16-rounded exact, and vs bf16 rounding
            print(f"c={c} d={d} fit_err={e:.2e} fp16eval max_abs={err.max():.2e} at y={y[err.argmax()]:.2f} rms={np.sqrt((err**2).mean()):.2e} maxcoef={np.abs(coef).max():.3g}")
