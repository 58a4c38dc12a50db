"""What a block costs a real-time caller (GPU box): per fxb_process_block_dev call on device-resident PCM - the host's time to
enqueue it, the sustained time per block when blocks are queued back to back, and call + sync per block.
    python tools/block_call_cost.py"""
import sys,time,os
sys.path[:0]=["fx8010-emulator-core_amd/python","oracle"]
import torch, numpy as np
import fx8010_amd as A, fx8010_programs as P
for name,n,S in (("config2",4096,32),("config2",4096,64),("config5",4096,64),("config2",64,32)):
    b=A.Batch(n,1,0); assert b.load_text(P.CONFIGS[name]())
    x=torch.from_numpy(P.stimulus(n,S)).cuda(); y=torch.empty_like(x)
    for _ in range(20): b.process_block_dev(x.data_ptr(),y.data_ptr(),S)
    b.sync()
    N=2000
    t0=time.perf_counter()
    for _ in range(N): b.process_block_dev(x.data_ptr(),y.data_ptr(),S)
    t1=time.perf_counter()
    b.sync()
    t2=time.perf_counter()
    # with a sync per block (a real-time caller)
    t3=time.perf_counter()
    for _ in range(500):
        b.process_block_dev(x.data_ptr(),y.data_ptr(),S); b.sync()
    t4=time.perf_counter()
    print("%s n=%d S=%d: enqueue %.1f us/call, sustained %.1f us/block (kernel %.1f us), call+sync %.1f us/block, stages %d"%(name,n,S,(t1-t0)/N*1e6,(t2-t0)/N*1e6,b.last_kernel_ms()*1e3,(t4-t3)/500*1e6,b.info("waves_per_wg")))
