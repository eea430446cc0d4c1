import numpy as np, sys
sys.path.insert(0,'.')
import imageprocessing_amd as cv2
from oracle import frontref as fr
h,w=127,129
g=np.random.default_rng(h*7+w).integers(0,256,(h,w),dtype=np.uint8)
for lo,hi in ((10,10),(10,400),(0,400),(10,300)):
    a=cv2.Canny(g,lo,hi); b,m=fr.canny(g,lo,hi,return_map=True)
    d=np.argwhere(a!=b)
    print(lo,hi,"ndiff",len(d),[(int(y),int(x),int(a[y,x]),int(b[y,x]),int(m[y,x])) for y,x in d[:10]])
b,m=fr.canny(g,10,400,return_map=True)
for (y,x) in ((3,128),(56,128)):
    print((y,x)); print(m[max(y-2,0):y+3, x-3:x+1]); print(b[max(y-2,0):y+3, x-3:x+1])
# same image but padded to 133 columns so x=128 is not the last column
g2=np.concatenate([g, g[:, -4:]], axis=1)
a=cv2.Canny(g2,10,400); b2=fr.canny(g2,10,400); print("w=133 ndiff", (a!=b2).sum())
g3=g[:, :128]
a=cv2.Canny(g3,10,400); b3=fr.canny(g3,10,400); print("w=128 ndiff", (a!=b3).sum())
