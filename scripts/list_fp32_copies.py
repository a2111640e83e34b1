import sys; sys.path.insert(0,'/root/repo')
import yolov3
y = yolov3.Yolo((416,416,3), [f"c{i}" for i in range(80)]); y.create_model(pretrained_body=None)
net = y.model.net
tot=0
for u in net.units:
    if u.kind=="conv" and u.bn and u.a_needed:
        mb = 32*u.out.h*u.out.w*u.cout*4/1e6
        tot+=mb
        print(u.name, u.out.h, u.cout, round(mb,1),"MB")
print("total fp32 a written per step (bs32): %.0f MB"%tot)
