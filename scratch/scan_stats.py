"""scratch/scan_stats.py -- how many samples does a LaserScanSensor beam of the config-5 workload march before its second hit,
per beam and per wave of 64 adjacent beams, and how many iterations would a clearance-grid (Chebyshev distance) march need?
(CPU replay on oracle states; numbers in profiles/r04_kernel_geometry.md section 3.)"""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, math
from oracle import ca_oracle as orc
import bench
from scipy import ndimage
table=np.load('/root/repo/gym_collision_avoidance_amd/data/test_cases.npz')['n50']
E=6; N=50
o=orc.Oracle(orc.default_params(E,N,max_obs=49)); o.set_policies(orc.POL_RVO)
grid=bench.crowd_map(); o.set_map(grid)
o.reset(table[:E])
for t in range(60): o.rollout_ex(table,1)
B=512; R=60
ang=np.linspace(-math.pi/2, math.pi/2, B)
res={}
for cap in (4,7,10,15):
    its=[]; wmax=[]; base=[]; wbase=[]
    for e in range(E):
        occ=grid.copy()
        px=o.view('pos_x')[e]; py=o.view('pos_y')[e]; rad=o.view('radius')[e]; hd=o.view('heading')[e]
        gr=np.floor(80 - py/0.1).astype(int); gc=np.floor(80+px/0.1).astype(int)
        rr,cc=np.mgrid[0:160,0:160]
        own=[]
        for a in range(N):
            m=((cc-gc[a])**2+(rr-gr[a])**2) < (rad[a]/0.1)**2
            own.append(m); occ|=m
        # chebyshev clearance
        d=ndimage.distance_transform_cdt(~occ, metric='chessboard')
        d=np.minimum(d,cap)
        dpad=np.pad(d,8,constant_values=cap); occpad=np.pad(occ,8)
        for a in range(N):
            ownpad=np.pad(own[a],8)
            n_it=np.zeros(B,int); n_base=np.zeros(B,int)
            for b in range(B):
                th=hd[a]+ang[b]; cs=math.cos(th); sn=math.sin(th)
                r=0; hits=0; it=0
                while r<R and hits<2:
                    x=px[a]+r*0.1*cs; y=py[a]+r*0.1*sn
                    g_r=int(math.floor(80-y/0.1))+8; g_c=int(math.floor(80+x/0.1))+8
                    it+=1
                    if g_r<0 or g_c<0 or g_r>=176 or g_c>=176:
                        break   # beyond the box (clip)
                    dd=dpad[g_r,g_c]
                    if dd==0 and not ownpad[g_r,g_c]:
                        hits+=1; r+=1
                    else:
                        r+=max(1,dd)
                n_it[b]=it
                # baseline: samples marched until second hit
                r=0; hits=0
                while r<R and hits<2:
                    x=px[a]+r*0.1*cs; y=py[a]+r*0.1*sn
                    g_r=int(math.floor(80-y/0.1))+8; g_c=int(math.floor(80+x/0.1))+8
                    if g_r<0 or g_c<0 or g_r>=176 or g_c>=176: break
                    if occpad[g_r,g_c] and not ownpad[g_r,g_c]: hits+=1
                    r+=1
                n_base[b]=r
            its.append(n_it.mean()); wmax.append(n_it.reshape(8,64).max(1).mean())
            base.append(n_base.mean()); wbase.append(n_base.reshape(8,64).max(1).mean())
        if e>=1 and cap!=7: break
    print("cap",cap,"iters/beam mean %.1f  wave-max mean %.1f | baseline samples mean %.1f wave-max %.1f"%(np.mean(its),np.mean(wmax),np.mean(base),np.mean(wbase)))
