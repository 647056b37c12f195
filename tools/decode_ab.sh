#!/bin/bash
# A/B of a -D switch of csrc/mlp_decode.hip on the GPU box: times the shipped build, rebuilds with the switch, times again
cd $GRAFT_REPO_ROOT
T='import sys,torch;sys.path.insert(0,".");import eps_amd;from eps_amd import ops
dev=torch.device("cuda:0");N,H,E=576289,256,1<<22;g=torch.Generator(device=dev).manual_seed(0)
x=torch.randn(N,H,generator=g,device=dev);u=torch.randint(0,N,(E,),generator=g,device=dev,dtype=torch.int32);v=torch.randint(0,N,(E,),generator=g,device=dev,dtype=torch.int32)
for nl in (2,3):
    ws=[torch.randn(H if i<nl-1 else 1,H,generator=g,device=dev)/16 for i in range(nl)];bs=[torch.randn(H if i<nl-1 else 1,generator=g,device=dev) for i in range(nl)]
    for _ in range(3): o=ops.mlp_decode(x,u,v,ws,bs)
    torch.cuda.synchronize();a=torch.cuda.Event(enable_timing=True);b=torch.cuda.Event(enable_timing=True);a.record()
    for _ in range(10): o=ops.mlp_decode(x,u,v,ws,bs)
    b.record();torch.cuda.synchronize();print("L=%d %.3f ms  checksum %.6f"%(nl,a.elapsed_time(b)/10,float(o.double().sum())))'
echo "shipped:"; python -c "$T"
touch edge-proposal-sets_amd/csrc/mlp_decode.hip
make -C edge-proposal-sets_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $1" 2>/dev/null
echo "with $1:"; python -c "$T"
