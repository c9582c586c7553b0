"""Build-level regression checks on the gfx950 ISA of the shipped kernels (CPU only: hipcc cross-compiles without a GPU).

Two hipcc (ROCm 7.2, clang 22) miscompiles are worked around in slam3d_gx_amd/csrc/icp_kernels.hpp; a compiler upgrade can
un-fix or re-break either silently, and the GPU parity suite would only notice as a hang or a wild load.  These tests look
at the generated code itself.  Each names the GPU parity test that exercises the construct.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam3d_gx_amd import build as B  # noqa: E402


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "dev.s"
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([B.hipcc()] + flags + ["--cuda-device-only", "-S", os.path.join(B.CSRC, "icp_capi.hip"), "-o", str(out)])
    return out.read_text()


def _kernel_bodies(asm):
    """mangled name -> text between its label and its .end_amdhsa_kernel / s_endpgm block"""
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.section\s+\.rodata", asm, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    return bodies


def _meta(asm):
    recs, cur = [], None
    for line in asm.splitlines():
        if re.match(r"  - \.\w+:", line):
            cur = {}
            recs.append(cur)
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip()
    return {r["name"]: r for r in recs if r.get("name", "").startswith("_Z")}


def _coop_production(names):
    # k_nn_tiles_acc<3, 7, true (COOP), false (DBG), false (GATED)>: Itanium mangling ...ILi3ELi7ELb1ELb0ELb0EE...
    return [n for n in names if "k_nn_tiles_acc" in n and "ILi3ELi7ELb1ELb0ELb0EE" in n]


def test_lds_fetch_add_uniform_stays_one_exec_masked_block(device_asm):
    """lds_fetch_add_uniform: `if (lane == 0) old = atomicAdd(p, v); old = readfirstlane(old)` inside the drain loops was
    restructured by hipcc so that lanes != 0 spun in an inner loop with lane 0 masked off (the cooperative build hung).  The
    fix is one opaque asm block: exec = lane 0, ds_add_rtn_u32, wait, restore.  It must reach the ISA as exactly that sequence
    in the cooperative kernel, once per call site (cell items, tile items, the two claim counters).
    Parity test of the construct: tests/test_gpu_parity.py::test_full_640x480_config2 (every launch drains through it)."""
    bodies = _kernel_bodies(device_asm)
    coop = _coop_production(bodies)
    assert len(coop) == 1, coop
    body = bodies[coop[0]]
    blocks = re.findall(r"s_mov_b64 (s\[\d+:\d+\]), exec\n\s*s_mov_b64 exec, 1\n\s*ds_add_rtn_u32 v\d+, v\d+, v\d+\n\s*s_waitcnt lgkmcnt\(0\)\n\s*s_mov_b64 exec, \1", body)
    assert len(blocks) >= 4, f"{len(blocks)} intact fetch-and-add blocks in the cooperative kernel"
    # no other LDS returning add except the two end-of-block stamp counters (stamp_wave_end: outside any loop): the fragile
    # source form of a drain-loop claim would show up as one more bare ds_add_rtn_u32
    assert body.count("ds_add_rtn_u32") - len(blocks) <= 2


def test_tile_item_mask_survives_in_front_of_the_record_address(device_asm):
    """A tile item is (owner << 24 | tile); the tile id is masked with 0xffffff before it is scaled to the record address (a
    64-bit multiply-add).  hipcc once dropped such a mask when the masked value also fed a 24-bit multiply (tools/attic,
    the cross-block sharing experiment): a wild load.  The mask must be in the ISA of the cooperative kernel's phase B.
    Parity test of the construct: tests/test_gpu_parity.py::test_randomised_configurations_stay_bit_identical."""
    bodies = _kernel_bodies(device_asm)
    body = bodies[_coop_production(bodies)[0]]
    assert re.search(r"[sv]_and_b32 \w+(\[\d+\])?, (0xffffff, \w+|\w+, 0xffffff)", body), "the 24-bit tile mask is gone"


def test_production_search_kernels_have_no_scratch_and_use_the_fp64_matrix_cores(device_asm):
    """The production instances of k_nn_tiles_acc (cooperative and throughput build) keep their register budget -- 72 / 64
    VGPRs at 7 / 8 waves per SIMD, no scratch --, and the epilogue's Gram accumulation is eight v_mfma_f64_16x16x4_f64 (spec
    S4, round 4); the full-scan mode's 64 v_mfma_f32_16x16x4_f32 and 64 v_mfma_f32_16x16x32_bf16 are there too.
    Parity: tests/test_gpu_parity.py::test_every_nn_mode_is_bit_identical."""
    meta = _meta(device_asm)
    bodies = _kernel_bodies(device_asm)
    prod = [n for n in meta if "k_nn_tiles_acc" in n and "Lb0ELb0EE" in n]         # <.., DBG = false, GATED = false>
    assert len(prod) == 2, prod
    for n in prod:
        assert int(meta[n]["private_segment_fixed_size"]) == 0 and int(meta[n]["vgpr_spill_count"]) == 0, (n, meta[n])
        assert int(meta[n]["vgpr_count"]) <= 72
        assert bodies[n].count("v_mfma_f64_16x16x4") == 8, bodies[n].count("v_mfma_f64_16x16x4")
    # round 5: EVERY instance of the search kernel -- the gated ones (spec S4g / S4p: what max_plane_residual2,
    # min_normal_cos and the plane-pair gate run) and the instrumented ones too -- is free of scratch; the gated cooperative
    # instance carried 12 B per lane until the source pixel was re-read in the epilogue instead of kept live across the drain
    every = [n for n in meta if "k_nn_tiles_acc" in n]
    assert len(every) == 6, every
    for n in every:
        assert int(meta[n]["private_segment_fixed_size"]) == 0 and int(meta[n]["vgpr_spill_count"]) == 0, (n, meta[n])
    mf = [n for n in bodies if "k_nn_mfmaE" in n]                                  # (Itanium names: 9k_nn_mfmaE... / 11k_nn_mfma16E...)
    assert len(mf) == 1 and bodies[mf[0]].count("v_mfma_f32_16x16x4") >= 64
    # the bf16-split form (round 4): 64+ v_mfma_f32_16x16x32_bf16 with the inline constant 0 as C, no scratch, and a loop head that
    # waits for ONE fragment (vmcnt(7)), not for all eight (tests/test_gpu_parity.py::test_matrix_core_scans_bf16_split_and_f32)
    mf16 = [n for n in bodies if "k_nn_mfma16E" in n]
    assert len(mf16) == 1 and bodies[mf16[0]].count("v_mfma_f32_16x16x32_bf16") >= 64
    assert int(meta[mf16[0]]["private_segment_fixed_size"]) == 0 and int(meta[mf16[0]]["vgpr_spill_count"]) == 0
    import re
    # (round 5: the scan loop is the OUTER loop now -- the drain of the flagged-group queue brings inner loops of its own --: the text from
    # its header to its first MFMA holds the wait for the fragment that MFMA consumes)
    outer = re.split(r"This Loop Header: Depth=1", bodies[mf16[0]])
    scan = [seg for seg in outer[1:] if "v_mfma_f32_16x16x32_bf16" in seg.split("Loop Header", 1)[0] or "v_mfma_f32_16x16x32_bf16" in seg[:6000]]
    assert scan, "no depth-1 loop with bf16 MFMAs"
    head = scan[0].split("v_mfma_f32_16x16x32_bf16", 1)[0]
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", head)
    assert waits and int(waits[-1]) >= 3, waits
    # and no wait inside the scan loop's own blocks drains everything before an MFMA: the flagged groups are queued (LDS), their
    # targets are loaded in the drain


def test_persistent_list_kernel_keeps_its_co_residency_budget(device_asm):
    """k_list_icp (point lists, round 6) meets its other blocks at a grid barrier, so every block of a launch must be resident:
    at most 256 blocks of four waves per launch, and four such launches (the runtime's four hardware queues) fit the chip only while a
    block needs <= 128 VGPRs, no scratch and <= 40 KB of LDS (four blocks per CU).  The production instances also must not fence:
    __threadfence() is buffer_wbl2 + buffer_inv on gfx950 -- an L2 write-back and invalidate per block and iteration.
    Parity: tests/test_unorganized.py."""
    meta = _meta(device_asm)
    bodies = _kernel_bodies(device_asm)
    inst = [n for n in meta if "k_list_icp" in n]
    assert len(inst) == 4, inst                                  # svd, planes, planes + gates, and the instrumented svd instance
    for n in inst:
        assert int(meta[n]["vgpr_count"]) <= 128 and int(meta[n]["private_segment_fixed_size"]) == 0 and int(meta[n]["vgpr_spill_count"]) == 0, (n, meta[n])
        assert int(meta[n]["group_segment_fixed_size"]) <= 40 * 1024, meta[n]["group_segment_fixed_size"]
        # (the device functions this kernel calls out of line sit in front of it and would swallow its label in _kernel_bodies)
        body = device_asm.split("\n" + n + ":", 1)[1].split("s_endpgm", 1)[0]
        assert "buffer_wbl2" not in body and "buffer_inv" not in body, "a device-scope fence crept into the persistent kernel"
        assert body.count("v_mfma_f64_16x16x4") == 8
        # the tile loads are issued by hand and waited for with vmcnt(2): three in flight
        assert len(re.findall(r"s_waitcnt vmcnt\(2\)", body)) >= 3


def test_voxel_insert_keeps_six_tile_blocks_per_cu(device_asm):
    """k_voxel_insert (row f-1, round 6): a frame alone is 1,200 tile blocks that must all be resident at once (the launch is one block
    lifetime long), and batches are bound by block lifetime x occupancy -- the tile instance stays under 160 KB / 6 of LDS, the list instance
    (1,024 threads, 2,048-entry table) within the 160 KB of a CU, no scratch in either; the scan and finalize launches use no fence (buffer_wbl2 / buffer_inv).
    Parity: tests/test_voxel.py."""
    meta = _meta(device_asm)
    tile = [n for n in meta if "k_voxel_insertILb1" in n]
    lst = [n for n in meta if "k_voxel_insertILb0" in n]
    assert len(tile) == 1 and len(lst) == 1, (tile, lst)
    assert int(meta[tile[0]]["group_segment_fixed_size"]) <= 160 * 1024 // 6, meta[tile[0]]["group_segment_fixed_size"]
    assert int(meta[lst[0]]["group_segment_fixed_size"]) <= 160 * 1024, meta[lst[0]]["group_segment_fixed_size"]
    for n in tile + lst + [m for m in meta if "k_voxel_scan" in m or "k_voxel_finalize" in m]:
        assert int(meta[n]["private_segment_fixed_size"]) == 0 and int(meta[n]["vgpr_spill_count"]) == 0, (n, meta[n])
        assert int(meta[n]["vgpr_count"]) <= 64, (n, meta[n]["vgpr_count"])
        body = device_asm.split("\n" + n + ":", 1)[1].split("s_endpgm", 1)[0]
        assert "buffer_wbl2" not in body and "buffer_inv" not in body, n
