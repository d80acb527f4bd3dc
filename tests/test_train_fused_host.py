"""CPU: host side of the fused training stack (diffsinger_amd/train_fused.py, include/dsf.h "Training, the FUSED residual stack") - which
DiffNets it covers, the workspace layout the C side reports, and argument checking of the entry points (no device work)."""
import ctypes as C

import pytest
import torch

import diffsinger_amd
from diffsinger_amd import _lib, hparams, train_fused


def _net(**over):
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    hparams.update(over)
    return diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)


def test_which_denoisers_take_the_fused_stack(monkeypatch):
    assert train_fused.supported(_net())
    assert train_fused.supported(_net(dilation_cycle_length=4))
    assert not train_fused.supported(_net(residual_channels=384))
    assert not train_fused.supported(_net(hidden_size=384))
    assert not train_fused.supported(_net(dilation_cycle_length=5))          # dilation 16 > the 8-frame halo
    assert not train_fused.supported(_net(residual_layers=33))
    net = _net(residual_channels=384)
    assert not net.fused() and _net().fused()
    monkeypatch.setenv('DSD_TRAIN_FUSED', '0')
    assert not train_fused.enabled()
    monkeypatch.delenv('DSD_TRAIN_FUSED')
    assert train_fused.enabled()


def test_workspace_layout_and_argument_checks():
    lib = _lib.load()
    train_fused._bind(lib)
    for B, T, L in ((8, 1024, 20), (2, 50, 3), (1, 5, 1)):
        TS = (T + 31) // 32 * 32
        ntiles = B * TS // 32
        n0, n1 = lib.dsf_stack_workspace_floats(B, T, L, 0), lib.dsf_stack_workspace_floats(B, T, L, 1)
        off = (C.c_int64 * 16)()
        assert lib.dsf_stack_offsets(B, T, L, 0, off, 16) == 0
        w1p, wcp, w2p, b1p, cp, X, Y, A, skip, bsum, cp_l, X_l, Y_l, A_l, total = list(off)[:15]
        assert total == n0 and 0 == w1p < wcp < w2p < b1p < cp < X < Y < A < skip < bsum < total
        assert cp_l == ntiles * 16384 and X_l == ntiles * 8192 and A_l == ntiles * 16384
        assert Y_l == B * 256 * (TS + 16)                                  # rows of the saved y carry 8 zero floats on both sides
        assert all(v % 4 == 0 for v in (cp, X, Y, A, skip))                # float4 alignment of every sub-buffer
        assert lib.dsf_stack_offsets(B, T, L, 1, off, 16) == 0
        assert list(off)[9] == n1 and list(off)[0] == 0
        # per frame and layer: conditioner projection 2 KiB + x 1 KiB + y 1 KiB + gate pre-activation 2 KiB (8 x 1024 frames x 20 layers: 1 GB)
        assert L * B * TS * 6 * 1024 <= n0 * 4 <= L * B * TS * 6 * 1024 + 200 * 2 ** 20
    assert lib.dsf_stack_workspace_floats(8, 1024, 33, 0) == -1 and lib.dsf_stack_workspace_floats(0, 10, 2, 0) == -1
    assert lib.dsf_wgrad2_workspace_floats(512, 256, 3) > 0 and lib.dsf_wgrad2_workspace_floats(500, 256, 3) == -1
    assert lib.dsf_wgrad2_workspace_floats(512, 256, 5) == -1
    # null arguments are refused before any device call
    assert lib.dsf_stack_forward(None, None, None, None, 1, 8, 1, None, None, None) != 0
    assert b'null' in lib.dsd_last_error()
    assert lib.dsf_conv1d_wgrad2(None, None, None, None, None, 1, 256, 512, 3, 1, 8, None) != 0


def test_stack_mode_switch_and_workspace_sizes():
    """dsf_set_stack_mode (include/dsf.h) replaces the environment switches of round 2: it validates its argument, and the workspace sizes do
    not depend on it (a size query and the backward call could disagree when the environment changed in between)."""
    lib = _lib.load()
    train_fused._bind(lib)
    B, T, L = 8, 1024, 20
    sizes = []
    for mode in (1, 0, 2, 1):
        train_fused.set_stack_mode(mode)
        sizes.append((lib.dsf_stack_workspace_floats(B, T, L, 0), lib.dsf_stack_workspace_floats(B, T, L, 1)))
    assert len(set(sizes)) == 1 and sizes[0][0] > 0 and sizes[0][1] > 0
    with pytest.raises(RuntimeError):
        train_fused.set_stack_mode(3)
    train_fused.set_stack_mode(1)


def test_fused_path_refuses_cpu_tensors():
    net = _net()
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=100, loss_type='l1',
                                          spec_min=[-6.0] * 80, spec_max=[0.0] * 80).train()
    with pytest.raises(RuntimeError):
        gd.p_losses(torch.zeros(1, 1, 80, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 256, 8))


def test_conditioner_projection_grouping_rule(monkeypatch):
    """Host logic of the k_condproj launch (csrc/dsd_kernels.hpp condproj_groups, round 6): grid.y = a multiple of the dilation cycle's period
    with equal layer counts (<= 10) per workgroup once the batch gives the chip two workgroups per CU; one layer per workgroup for small
    batches, aperiodic dilations and under DSD_CP_GROUPS=layer."""
    lib = _lib.load()
    monkeypatch.delenv('DSD_CP_GROUPS', raising=False)

    def groups(dil, ntiles):
        lds = C.c_int64(0)
        g = lib.dsd_debug_condproj_groups(bytes(dil), len(dil), ntiles, C.byref(lds))
        assert lds.value == 256 * 32 * 4 + -(-len(dil) // g) * 2048 and lds.value <= 64 * 1024
        return g
    cyc4, cyc1 = [1, 2, 4, 8] * 5, [1] * 20
    assert groups(cyc4, 256) == 4 and groups(cyc4, 128) == 4 and groups(cyc4, 1024) == 4        # BASELINE configs[1] / [3]: 5 layers per workgroup
    assert groups(cyc4, 127) == 20 and groups(cyc4, 16) == 20                                  # 1 x 512: as many workgroups as there are
    assert groups(cyc1, 256) == 2 and groups(cyc1, 512) == 2                                   # never more than 10 layers per workgroup
    assert groups(cyc1, 255) == 4 and groups(cyc1, 100) == 10 and groups(cyc1, 40) == 20
    assert groups([1, 2, 4, 8, 1, 2, 4], 1024) == 7                                            # 7 layers: no multiple of the period divides them
    assert groups([1, 2, 4, 8, 2, 1, 4, 8], 4096) == 8                                         # aperiodic
    assert groups([1, 2] * 10, 256) == 2 and groups([1, 2] * 3, 256) == 2
    assert groups([4], 4096) == 1
    monkeypatch.setenv('DSD_CP_GROUPS', 'layer')
    assert groups(cyc4, 256) == 20
    monkeypatch.setenv('DSD_CP_GROUPS', '10')
    assert groups(cyc1, 16) == 10 and groups(cyc4, 256) == 20                                  # 10 is no multiple of the period 4: refused
    assert lib.dsd_debug_condproj_groups(None, 20, 256, None) == -1
