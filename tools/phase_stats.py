# event counters and wave phase statistics of C3 renders with 0 / 1 / 4 bounces (counting renders)
import sys; sys.path.insert(0,'.')
from gdpathtracing_amd import capi, host, scenes
sc=scenes.demo_scene(51200); W,H=1920,1080
for bounces in (0,1,4):
    ctx=host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(W,H,bounces,0); ctx.set_camera(scenes.camera_block(sc.camera,W,H))
    ctx.render(8,1,counted=True); st=ctx.stats(); p=st['phase']
    print("bounces",bounces,"rays",st['rays'],"blas",st['blas_expand'],"tlas",st['tlas_expand'],"tri",st['tri_tests'],"inst",st['inst_visits'])
    print("   rounds",p[0],"node_iters",p[1],"lanes/iter %.1f"%(p[2]/max(p[1],1)),"leaf_phases",p[3],"lanes/leafphase %.1f"%(p[4]/max(p[3],1)),"inst_phases",p[5],"lanes/instphase %.1f"%(p[6]/max(p[5],1)))
    ctx.close()
