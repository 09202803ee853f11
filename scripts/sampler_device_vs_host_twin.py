import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scripts')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis
side, n, noise, nlive = 8, 512, 0.1, 400
n_pix = side*side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side)); r = np.hypot(lon - side/2, lat - side/2)/(side/2)
truths = np.zeros((n_pix, 6)); truths[:,0] = -1+2*lon.ravel()/side; truths[:,1]=12; truths[:,2]=5; truths[:,3]=14.6-0.6*r.ravel(); truths[:,4]=0.4
probe = CubeRunner(axes,(1,2),np.zeros((1,2*n)),np.full((1,2),noise),ut,ncomp=1)
model,_ = probe.predict_batch(np.zeros(n_pix,dtype=np.int32), truths)
cube = CubeRunner(axes,(1,2),model+rng.normal(0,noise,model.shape),np.full((n_pix,2),noise),ut,ncomp=1)
for mode in ('fast','table'):
    na.set_exp_mode(mode)
    a = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, device=True, method='reject')
    b = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, device=False, method='reject')
    ea, eb = np.array([x.n_evals for x in a]), np.array([x.n_evals for x in b])
    print(mode, 'pixels with different evaluation counts:', int((ea != eb).sum()), 'of', n_pix, ' max |dlnZ|', max(abs(x.lnZ-y.lnZ) for x,y in zip(a,b)))
