# Extended fuzz on the GPU box (64 more seeds than tests/test_fuzz.py, varying sizes, bounces, frames, modes and
# asynchronous renders): every reference-tree route bit for bit against the oracle, the native tree against the
# tree-independent oracle mode.   gpurun -- python tools/fuzz_more.py
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from gdpathtracing_amd import capi, host, scenes, wire
from oracle import binding as ob
bad=0
import faulthandler; faulthandler.enable()
import os
for seed in range(int(os.environ.get('FUZZ_FROM', '8')), int(os.environ.get('FUZZ_TO', '72'))):
    print("seed", seed, flush=True)
    sc = scenes.random_scene(seed, n_meshes=2+seed%4, n_instances=3+seed%11, tris_per_surface=9+7*(seed%9), coincident=(seed%3!=0))
    w,h,b,f = 96+8*(seed%5), 64+8*(seed%3), seed%5, 1+seed%3
    cam = scenes.camera_block(sc.camera,w,h); ref = ob.build_scene(sc)
    mode = seed%2
    want,wl,wd,cnt,_ = ob.render(ref,cam,w,h,b,f,1+seed,mode)
    for route in ("upload","exact"):
        for kernel in (capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT):
            print("  ", route, kernel, flush=True); ctx = host.Context(0); ctx.set_kernel(kernel)
            if route=="upload": ctx.upload_reference_layout(ref.tri_geom,ref.tri_data,ref.materials,ref.bvh_nodes,ref.instances,ref.tlas_nodes,ref.textures,as_given=True)   # node for node: ties too
            else: ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT)
            ctx.set_params(w,h,b,mode); ctx.set_camera(cam)
            ctx.render(f,1+seed,asynchronous=(seed%4==1)); 
            ok = np.array_equal(ctx.read_accum(),want,equal_nan=True) and np.array_equal(ctx.read_ldr(),wl) and np.array_equal(ctx.read_depth(),wd,equal_nan=True)
            if not ok: bad+=1; print("MISMATCH seed",seed,route,kernel)
            ctx.close()
    if seed%3==0:
        # native tree with reach records against the oracle's reach-only mode (every triangle tested + the two reach tests),
        # native tree alone against the tree-independent mode; in both accumulation modes, blocking and queued
        for builder, flags in ((capi.BUILD_SAH, 2), (capi.BUILD_SAH_WATERTIGHT, 1), ("upload", 2)):   # "upload": route (i) on the native tree
            amode = (seed // 3) % 2
            wantn,wln,wdn,_,_ = ob.render(ref,cam,w,h,b,f,1+seed,amode,flags=flags)
            ctx = host.Context(0)
            if builder == "upload":
                ctx.upload_reference_layout(ref.tri_geom,ref.tri_data,ref.materials,ref.bvh_nodes,ref.instances,ref.tlas_nodes,ref.textures)
                assert ctx.tree_kind() == capi.TREE_NATIVE_REACH, ctx.upload_note()
            else: ctx.build_scene(sc, builder)
            ctx.set_params(w,h,b,amode); ctx.set_camera(cam)
            ctx.render(f,1+seed,asynchronous=(seed%2==1)); ctx.sync()
            got=ctx.read_accum(); gd=ctx.read_depth(); ctx.close()
            m = ~(np.isnan(got).any(-1)|np.isnan(wantn).any(-1))
            if not (np.array_equal(np.isnan(got).any(-1),np.isnan(wantn).any(-1)) and np.array_equal(got[m],wantn[m]) and np.array_equal(gd,wdn,equal_nan=True)):
                bad+=1; print("SAH MISMATCH seed",seed,"builder",builder, int((got[m]!=wantn[m]).any(-1).sum()))
print("extended fuzz done, mismatches:",bad)
