"""tools/scattered_taylor_prototype.py -- numerical prototype (numpy, float64) for the scattered model: the weights of the seven
stencil points of a sample from ONE second-order expansion about the centre (w, dw/dr, d2w/dr2, dw/dh, d2w/dr dh with
r_g - r_c = -o.u + (|o|^2 - (o.u)^2)/(2 r)) against the exact weights, on BASELINE config[4] sample set: error of ln N at the
seven points and of its central-difference gradient.  Result (150 stencils): ln N abs error median 7e-14, max 4e-13; gradient
relative error median 3e-8, p90 1.2e-7, max 2.5e-6 (bars of the parity tests: median 1e-5, p90 1e-3).  HISTORY section 12.1."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from stanford_raytracer_amd import workloads as wl
from scipy.spatial import cKDTree
np.seterr(all='ignore')
pts, lnN = wl.make_points_config5(5)
tree = cKDTree(pts)
d2nn,_ = tree.query(pts, k=2, workers=8); nn = d2nn[:,1]
rr = np.linalg.norm(pts,axis=1); nn[rr<wl.R_E]=1.0
R = 1.5*nn[rr>=wl.R_E].max(); lws=5.0; eps0=5e-16
def window(r): return 0.5+0.5*np.cos(np.pi*r/R)
def eta(r,h):
    x=(r+R*eps0)/(h/4); return np.exp(-x**1.1)*window(r)
def monos(dx):
    x,y,z=dx[:,0],dx[:,1],dx[:,2]
    return np.stack([np.ones_like(x),z,z*z,y,y*z,y*y,x,x*z,x*y,x*x],axis=1)
def fit_exact(p, idx):
    q=pts[idx]; dx=q-p; r=np.linalg.norm(dx,axis=1); m=r<R
    q,dx,r,ii=q[m],dx[m],r[m],idx[m]
    cw=window(r); h=lws*(cw*nn[ii]).sum()/cw.sum()
    w=0.5*eta(r,h)
    E=monos(dx); A=(E*w[:,None]).T@E; b=(E*w[:,None]).T@lnN[ii]
    y=np.linalg.solve(A,np.eye(10)[0]); return y@b, h
pos,dirs,ws = wl.launch_set(400,5)
errs=[];graderr=[]
for p in pos[:150]:
    if np.linalg.norm(p)<wl.R_E*1.05: continue
    idx=np.array(tree.query_ball_point(p,R*1.001))
    if len(idx)<30: continue
    d=1e-6*np.abs(p); d=np.maximum(d,1e-6)
    offs=[np.zeros(3)]
    for a in range(3):
        e=np.zeros(3); e[a]=d[a]; offs+= [e,-e]
    exact=[fit_exact(p+o,idx) for o in offs]
    vals_exact=np.array([e[0] for e in exact]); h_exact=np.array([e[1] for e in exact])
    # ---- Taylor basis about the centre
    q=pts[idx]; dx=q-p; r=np.linalg.norm(dx,axis=1); m=r<R
    q,dx,r,ii=q[m],dx[m],r[m],idx[m]; u=dx/r[:,None]
    f=window(r); fp=-0.5*np.pi/R*np.sin(np.pi*r/R); fpp=-0.5*(np.pi/R)**2*np.cos(np.pi*r/R)
    # h_g to second order in o:  r_g = |dx - o| ~ r - o.u + (|o|^2-(o.u)^2)/(2r)
    def h_of(o):
        ou=u@o; dr=-ou+(o@o-ou*ou)/(2*r)
        cw=f+fp*dr+0.5*fpp*dr*dr
        return lws*(cw*nn[ii]).sum()/cw.sum()
    hc=h_of(np.zeros(3))
    # weights: W(r,h)=0.5*exp(-x^1.1)*f(r), x=(r+R eps)/(h/4)
    def W(r_,h_): return 0.5*eta(r_,h_)
    x=(r+R*eps0)/(hc/4); U=x**1.1; Ew=np.exp(-U)
    dlnE_dr=-1.1*U/(r+R*eps0); d2lnE_dr2=-1.1*0.1*U/(r+R*eps0)**2
    w0=0.5*Ew*f
    w_r=0.5*Ew*(dlnE_dr*f+fp)
    w_rr=0.5*Ew*((d2lnE_dr2+dlnE_dr**2)*f+2*dlnE_dr*fp+fpp)
    w_h=0.5*Ew*f*(1.1*U/hc)            # d/dh of exp(-(r/(h/4))^1.1) = E*1.1*U/h
    w_rh=0.5*(Ew*(1.1*U/hc))*(dlnE_dr*f+fp)+0.5*Ew*f*(1.1*1.1*U/hc/(r+R*eps0))
    E0=monos(dx)
    vals=[]
    for o in offs:
        ou=u@o; dr=-ou+(o@o-ou*ou)/(2*r); dh=h_of(o)-hc
        wg=w0+w_r*dr+0.5*w_rr*dr*dr+w_h*dh+w_rh*dr*dh
        # basis about the centre, fitted value at offset o
        A=(E0*wg[:,None]).T@E0; b=(E0*wg[:,None]).T@lnN[ii]
        mo=monos((-o)[None,:])[0]      # monomials of (p_g - c) with dx convention q-p: value at p_g = m(p_g - c)... sign: dx = q - c ; point offset = o -> local coord of p_g is +o
        mo=monos(o[None,:])[0]
        y=np.linalg.solve(A,mo); vals.append(y@b)
    vals=np.array(vals)
    errs.append(np.abs(vals-vals_exact).max())
    # gradient error: central differences
    ge=[];gx=[]
    for a in range(3):
        ge.append((vals_exact[1+2*a]-vals_exact[2+2*a])/(2*d[a])); gx.append((vals[1+2*a]-vals[2+2*a])/(2*d[a]))
    ge=np.array(ge);gx=np.array(gx)
    graderr.append(np.linalg.norm(gx-ge,axis=0).max()/max(np.linalg.norm(ge,axis=0).max(),1e-300))
errs=np.array(errs);graderr=np.array(graderr)
print('stencils',len(errs)); print('abs err of lnN (max over 7 points, 4 species): median %.3g p90 %.3g max %.3g'%(np.median(errs),np.percentile(errs,90),errs.max()))
print('relative error of the central-difference gradient of lnN: median %.3g p90 %.3g max %.3g'%(np.median(graderr),np.percentile(graderr,90),graderr.max()))
