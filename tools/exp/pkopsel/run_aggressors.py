"""pk_opsel_kernel (mode 0: v_pk_add_f32 op_sel:[0,1]) beside single-instruction aggressor loops."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device('cuda:0')
g = ctypes.CDLL(os.path.join(here, 'libpkopsel.so'))
g.pk_opsel_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
a = ctypes.CDLL(os.path.join(here, 'libaggressor.so'))
a.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
inp = torch.rand(1 << 20, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
out = torch.zeros(16, device=dev)
side = torch.cuda.Stream()
kinds = ['v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x2_f32', 'v_cvt_pk_bf16_f32', 'v_perm_b32', 'LDS b64 write+read', 'v_mfma_f32_16x16x32_bf16', 'v_pk_fma_f32', 'v_mfma_f32_32x32x8_f16', 'v_pk_mov_b32 op_sel:[1,0]', 'v_pk_mov_b32', 'v_pk_mul_f32', 'v_mov_b64', 'v_pk_mov_b32 op_sel:[0,1]', 'v_pk_mov_b32 op_sel_hi:[0,1]', 'waves: mfma | pk_mov op_sel', 'waves: mfma | pk_mul', 'waves: mfma | cvt_pk_bf16', 'mfma + pk_mov/pk_mul', 'mfma + cvt_pk_bf16', 'mfma + LDS', '2 independent mfma bf16', '2 independent mfma f32', 'cvt_pk + 2 mfma bf16']
iters = [60000, 30000, 400000, 400000, 100000, 100000, 400000, 60000] + [200000] * 6 + [60000] * 6 + [40000, 20000, 40000]
for victim in (0, 1):
    for k, name in enumerate(kinds):
        if k < 20 and not os.environ.get('ALL'): continue
        bad.zero_()
        torch.cuda.synchronize()
        e0, e1, f0, f1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e0.record()
            assert g.pk_opsel_launch(inp.data_ptr(), bad.data_ptr(), 2048, 20000, victim, side.cuda_stream) == 0
            e1.record()
        f0.record()
        assert a.aggressor_launch(out.data_ptr(), 2048, iters[k], k, torch.cuda.current_stream().cuda_stream) == 0
        f1.record()
        torch.cuda.synchronize()
        print('victim mode %d beside %-26s mismatching results %9d  (victim %.1f ms, aggressor %.1f ms)' % (victim, name, int(bad.item()), e0.elapsed_time(e1), f0.elapsed_time(f1)), flush=True)
