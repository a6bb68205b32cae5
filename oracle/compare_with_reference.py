"""TEST INFRASTRUCTURE ONLY. Ad-hoc sweep: oracle restatement vs oracle/_ref/ref_driver (unmodified reference) on synthetic and edge images.
The curated subset lives in tests/test_oracle_vs_reference.py."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from yaik_amd.synth import synth_planes
from oracle.refrun import run_reference
from oracle.pyoracle import OracleEncoder, OracleDecoder, PASSES, palette_remap, palette_decompress

def compare(planes, tag):
    n,h,w = planes.shape
    ref = run_reference(planes)
    enc = OracleEncoder(planes)
    bad = []
    def chk(name, a, b):
        a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
        ok = a.shape == b.shape and np.array_equal(a, b)
        if not ok:
            bad.append(name)
            print('  MISMATCH', name, a.shape, b.shape, (np.nonzero(a[:min(len(a),len(b))]!=b[:min(len(a),len(b))])[0][:5] if len(a) and len(b) else ''))
    if n == 4:
        m = enc.mip_prefilter()
        rb = np.frombuffer(ref['mip_bounds'], np.int32)
        chk('mip_bounds', m['bounds'], rb[:4]); chk('mip_remaining', [m['remaining']], rb[5:6])
        chk('mip_mask', enc.state('mipmapMask'), np.frombuffer(ref['mip_mask'], np.uint8))
        chunk = ref['mip_chunk']
        if m['has_chunk']:
            # HeaderBase(8) + MipmapHeader(bbox 8, streamSize 4, version 1, level 1, pad 2 = 16) + bitmap
            bm = np.frombuffer(chunk[24:24+m['bitmap'].size], np.uint8)
            chk('mip_bitmap', m['bitmap'], bm)
            chk('mip_tilebbox', m['tile_bbox'], np.frombuffer(chunk[8:16], np.int16))
        else:
            chk('mip_nochunk', [len(chunk)], [0])
    counts = np.frombuffer(ref['grad_counts'], np.int32)
    streams = []
    for i,(sx,sy) in enumerate(PASSES):
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        chk(f'count{i}', [cnt], [counts[i]])
        chk(f'bitmap{i}', bm, np.frombuffer(ref[f'grad_bitmap_{i}'], np.uint8))
        refdq = np.frombuffer(ref[f'grad_rgbdq_{i}'], np.uint8)
        if cnt:
            pal = enc.palette_compress(rgb)
            chk(f'palette{i}', pal, np.frombuffer(ref[f'grad_palette_{i}'], np.uint8))
            chk(f'paldec{i}', palette_decompress(pal, rgb.size), refdq)
            if not np.array_equal(palette_remap(rgb,250), refdq): print('   note: reference palette codec corrupts pass', i, int((palette_remap(rgb,250)!=refdq).sum()), 'bytes')
        streams.append((bm, refdq))
    chk('smoothMap', enc.state('smoothMap'), np.frombuffer(ref['smoothMap'], np.uint8))
    chk('mipmapMask_post', enc.state('mipmapMask'), np.frombuffer(ref['mipmapMask_post'], np.uint8))
    for p in range(3):
        chk(f'mapSmoothTile{p}', enc.state('mapSmoothTile',p), np.frombuffer(ref[f'mapSmoothTile_{p}'], np.uint8))
        chk(f'preview{p}', enc.state('preview',p), np.frombuffer(ref[f'preview_{p}'], np.int16))
    for m in range(2):
        for p in range(3):
            defs, nib, nn, dst = enc.dynamic_tile_encode(p, bool(m))
            chk(f'defs_{m}_{p}', defs, np.frombuffer(ref[f'plnt_defs_{m}_{p}'], np.uint16))
            chk(f'idx_{m}_{p}', nib, np.frombuffer(ref[f'plnt_idx_{m}_{p}'], np.uint8))
            chk(f'dst_{m}_{p}', dst, np.frombuffer(ref[f'plnt_dst_{m}_{p}'], np.int16))
    for p in range(3):
        tiles, dbg = enc.dynamic_tile_compressor(p)
        chk(f'd1_out{p}', dbg, np.frombuffer(ref[f'd1_out_{p}'], np.int16))
    pix, typ = enc.streams_1d()
    chk('d1_pix', pix, np.frombuffer(ref['d1_pix'], np.uint8)); chk('d1_type', typ, np.frombuffer(ref['d1_type'], np.uint8))
    dec = OracleDecoder(w,h)
    for i,(sx,sy) in enumerate(PASSES):
        if counts[i]: dec.gradient(sx, sy, *streams[i])
    chk('dec_planes_grad', dec.planes(), np.frombuffer(ref['dec_planes_grad'], np.uint8))
    chk('dec_tile4x4', dec.tile4x4(), np.frombuffer(ref['dec_tile4x4'], np.uint8))
    chk('dec_mapRGBMask', dec.map_rgb_mask(), np.frombuffer(ref['dec_mapRGBMask'], np.uint8))
    # mapRGB only meaningful where mask set
    dec.split_masks()
    tp, pp = dec.decode_1d(typ, pix)
    chk('dec_1d_consumed', [tp,pp], np.frombuffer(ref['dec_1d_consumed'], np.int32))
    chk('dec_planes_full', dec.planes(), np.frombuffer(ref['dec_planes_full'], np.uint8))
    print(tag, 'counts', counts, 'OK' if not bad else f'BAD {bad}')
    return not bad

ok = True
for W in (64, 128, 256, 512):
  for npl in ((3,) if W < 256 else (3,4)):
    ok &= compare(synth_planes(W, n_planes=npl), f"synth{W}x{npl}")
for W in ():
    for npl in (3,4):
        ok &= compare(synth_planes(W, n_planes=npl), f'synth{W}x{npl}')
print('ALL OK' if ok else 'FAIL')

rng = np.random.default_rng(7)
def mk(w,h,kind,npl=3):
    y,x = np.mgrid[0:h,0:w]
    if kind=='flat': rgb = np.stack([np.full((h,w),200),np.full((h,w),17),np.full((h,w),90)])
    elif kind=='noise': rgb = rng.integers(0,256,(3,h,w))
    elif kind=='ramp': rgb = np.stack([(x*2)%256,(y*3)%256,((x+y))%256])
    elif kind=='smooth': rgb = np.stack([(x*255)//w,(y*255)//h,((x+y)*255)//(w+h)]) + rng.integers(0,3,(3,h,w))
    elif kind=='white': rgb = np.full((3,h,w),255); rgb[:, h//2:, :] = rng.integers(250,256,(3,h-h//2,w))
    elif kind=='mixed':
        rgb = np.stack([(x*255)//w,(y*255)//h,((x+y)*255)//(w+h)])
        m = ((x//16 + y//16) % 3)==0
        rgb = np.where(m, rng.integers(0,256,(3,h,w)), rgb)
        m2 = ((x//8 + y//8) % 5)==0
        rgb = np.where(m2, np.clip(rgb + rng.integers(-6,7,(3,h,w)),0,255), rgb)
    rgb = np.clip(rgb,0,255)
    pl = [rgb[0],rgb[1],rgb[2]]
    if npl==4:
        a = np.full((h,w),255); a[:, :32]=0; a[:16,:]=0
        a[(x//16%4==1)&(y//16%3==1)] = 0
        a[40:56, 40:57] = np.where(rng.integers(0,4,(16,17))==0, 7, 0)
        pl.append(a)
    return np.ascontiguousarray(np.stack(pl).astype(np.int32))
for (w,h) in ((64,64),(128,128),(72,40),(200,136),(256,256)):
    for kind in ('flat','noise','ramp','smooth','mixed','white'):
        for npl in ((3,4) if (w==h and w in (128,256)) else (3,)):
            try:
                ok &= compare(mk(w,h,kind,npl), f'{kind}{w}x{h}x{npl}')
            except Exception as e:
                print('EXC', kind,w,h,npl, repr(e)[:200]); ok=False
print('ALL OK' if ok else 'FAIL')
