# Extended fuzz on the GPU box (64 more seeds than tests/test_fuzz.py, varying sizes, bounces, frames, modes and
# asynchronous renders): every reference-tree route AND both native routes bit for bit against the oracle's walk of the
# reference tree, the watertight builder against the tree-independent oracle mode.   gpurun -- python tests/tools/fuzz_more.py
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes, wire
from oracle import binding as ob
bad=0; n_ref=0; n_native=0; n_aside=0
import faulthandler; faulthandler.enable()
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
            n_ref+=1
            if not ok: bad+=1; print("MISMATCH seed",seed,route,kernel)
            ctx.close()
    # The native routes (reach records + the tie walk on the reference's own trees) against the oracle's NORMAL walk of the
    # reference tree, on every seed -- two thirds of them with coincident triangles, whose winner is a matter of visiting order;
    # the watertight builder against the tree-independent mode on the seeds without coincident triangles
    for builder, flags in ((capi.BUILD_SAH, 0), ("upload", 0), (capi.BUILD_SAH_WATERTIGHT, 1)):
      if flags == 1 and seed % 3 != 0: continue
      amode = (seed // 3) % 2
      wantn,wln,wdn,_,_ = (want,wl,wd,None,None) if (flags == 0 and amode == mode) else ob.render(ref,cam,w,h,b,f,1+seed,amode,flags=flags)
      for kernel in (capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT):   # (the audit kernel decides its ties the same way)
        ctx = host.Context(0); ctx.set_kernel(kernel)
        if builder == "upload":
            ctx.upload_reference_layout(ref.tri_geom,ref.tri_data,ref.materials,ref.bvh_nodes,ref.instances,ref.tlas_nodes,ref.textures)
            assert ctx.tree_kind() == capi.TREE_NATIVE_REACH, ctx.upload_note()
        else: ctx.build_scene(sc, builder)
        ctx.set_params(w,h,b,amode); ctx.set_camera(cam)
        ctx.render(f,1+seed,asynchronous=(seed%2==1)); ctx.sync()
        got=ctx.read_accum(); gd=ctx.read_depth(); st=ctx.stats(); ctx.close(); n_native+=1; n_aside+=st["set_aside"]
        m = ~(np.isnan(got).any(-1)|np.isnan(wantn).any(-1))
        if not (np.array_equal(np.isnan(got).any(-1),np.isnan(wantn).any(-1)) and np.array_equal(got[m],wantn[m]) and np.array_equal(gd,wdn,equal_nan=True)):
            bad+=1; print("NATIVE MISMATCH seed",seed,"builder",builder,"kernel",kernel,"pixels",int((got[m]!=wantn[m]).any(-1).sum()),"set aside",st["set_aside"],"dropped",st["set_aside_dropped"])
print("extended fuzz done: %d reference-tree renders, %d native-tree renders (%d vertices set aside: cracks and exact ties), mismatches: %d" % (n_ref, n_native, n_aside, bad))
