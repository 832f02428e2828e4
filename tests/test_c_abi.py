"""The C-ABI library (include/ss_hip.h): it loads without a GPU, exports every declared symbol, and rejects bad
arguments with SS_EINVAL before touching the device.  No compute calls here (CPU container)."""
import ctypes
import os
import re

import pytest

from ss_amd import _lib, planning

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ss_hip.h")).read()
    return sorted(set(re.findall(r"\bint\s+(ss_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert set(names) == set(_lib.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None


def test_geometry_constants_match_python_planning():
    lib = _lib.load()
    assert lib.ss_block_len() == planning.KB
    assert lib.ss_spec_floats() == planning.SPEC_FLOATS
    assert lib.ss_version() >= 1


def test_argument_checks_return_einval_without_a_gpu():
    lib = _lib.load()
    null = None
    one = ctypes.c_void_p(16)          # non-null dummy pointer: never dereferenced on these paths
    assert lib.ss_source_windows_f32(null, one, one, 4, null) == -1
    assert lib.ss_source_windows_f32(one, one, one, 0, null) == 0                        # empty batch is a no-op
    assert lib.ss_fftconv_binaural_f32(one, one, one, one, null, 2, 32000, 16000, 1, 16000, 16000, 16000, 0, null) == -1
    assert lib.ss_fftconv_binaural_f32(one, one, one, one, one, 2, 32000, 16000, 1, 16000, 16001, 16000, 0, null) == -1   # n_valid > out_len
    assert lib.ss_fftconv_binaural_f32(one, one, one, one, one, 2, 32000, 16000, 0, 16000, 16000, 16000, 0, null) == -1   # elem stride 0
    assert lib.ss_fftconv_binaural_f32(one, one, one, one, one, 0, 32000, 16000, 1, 16000, 16000, 16000, 0, null) == 0
    assert lib.ss_spectrogram_f32(one, one, 1, 100, 0, null) == -1                       # shorter than the reflect pad
    assert lib.ss_spectrogram_f32(one, one, 1, 16000, 7, null) == -1                     # unknown pad mode
    assert lib.ss_audio_obs_f32(one, one, one, one, null, null, 1, 32000, 16000, 1, 16000, 16000, 16000, 0, 0, null) == -1
    assert lib.ss_intensity_f32(one, one, 1, 16000, 0, null) == -1
    assert lib.ss_intensity_f32(null, one, 1, 16000, 150, null) == -1
    f = ctypes.c_float
    assert lib.ss_logmel_f32(one, one, 1, 16000, 0, one, one, 64, 23, f(1e-6), null) == -1          # max_len % 4
    assert lib.ss_logmel_f32(one, one, 1, 16000, 0, one, one, 129, 24, f(1e-6), null) == -1         # too many bands
    assert lib.ss_logmel_f32(one, one, 1, 16000, 0, one, one, 128, 36, f(1e-6), null) == -1         # table > 4096
    assert lib.ss_logmel_f32(one, one, 1, 16000, 0, one, one, 64, 24, f(0.0), null) == -1           # eps must be > 0
    assert lib.ss_logmel_f32(one, one, 1, 16000, 0, one, ctypes.c_void_p(20), 64, 24, f(1e-6), null) == -1   # unaligned table
    assert lib.ss_logmel_f32(one, one, 0, 16000, 0, one, one, 64, 24, f(1e-6), null) == 0           # empty batch
    assert lib.ss_gccphat_f32(one, one, 1, 16000, 0, 33, f(1e-8), null) == -1                       # max_lag > 32
    assert lib.ss_gccphat_f32(one, one, 1, 16000, 0, 0, f(1e-8), null) == -1
    assert lib.ss_gccphat_f32(one, one, 1, 100, 0, 8, f(1e-8), null) == -1                          # too short
    assert lib.ss_gccphat_f32(one, null, 1, 16000, 0, 8, f(1e-8), null) == -1


def test_python_layer_refuses_cpu_tensors():
    import torch
    from ss_amd import ops
    with pytest.raises(_lib.SsHipError):
        ops.spectrogram(torch.zeros((1, 2, 16000)))
    with pytest.raises(_lib.SsHipError):
        ops.intensity(torch.zeros((1, 2, 16000)))


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/ss_hip.h must compile as C99 (no C++-isms, no HIP/torch types) and a C
    translation unit must link against the library's symbols."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_abi.c"
    src.write_text(
        '#include "ss_hip.h"\n'
        'int main(void) {\n'
        '  /* argument checks only: no device needed */\n'
        '  if (ss_block_len() != 16384 || ss_spec_floats() != 32768) return 1;\n'
        '  if (ss_spectrogram_f32((const float*)16, (float*)16, 1, 100, SS_PAD_REFLECT, 0) != SS_EINVAL) return 2;\n'
        '  if (ss_gccphat_f32((const float*)16, (float*)16, 1, 16000, SS_PAD_CONSTANT, 99, 1e-8f, 0) != SS_EINVAL) return 3;\n'
        '  if (ss_fftconv_binaural_f32(0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, SS_FLAG_NO_DISTRACTOR, 0) != 0) return 4;\n'
        '  return 0;\n'
        '}\n')
    exe = tmp_path / "use_abi"
    so_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", so_dir, "-lss_hip", "-Wl,-rpath," + so_dir])
    assert subprocess.run([str(exe)]).returncode == 0


def _build_client(tmp_path, with_hip):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "ctx_client"
    so_dir = os.path.dirname(_lib.SO_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "ctx_client.c"),
           "-o", str(exe), "-L", so_dir, "-lss_hip", "-Wl,-rpath," + so_dir]
    if with_hip:
        cmd += ["-DWITH_HIP", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                "-Wno-error=unused-function", "-Wno-error=unused-parameter"]
    subprocess.check_call(cmd)
    return exe


def test_context_api_from_a_c_program_planning(tmp_path):
    """the self-contained entry used from C.  Host half (no GPU): a C99 client creates a context,
    registers two clips by length, plans two steps; the descriptors it prints equal the Python planner's, and the
    second step is all cache hits."""
    import subprocess
    exe = _build_client(tmp_path, with_hip=False)
    out = subprocess.run([str(exe), "plan"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "step 0 flags 0 new_windows 2"                # keys (sound 0, t0 0) [also the distractor's] and (1, 32000)
    assert lines[4] == "step 1 flags 0 new_windows 0"
    d = [[int(v) for v in l.replace("|", " ").split()[2:]] for l in lines[1:4]]
    ws0 = planning.plan_window_set(16000, 0, 1, 1)
    ws1 = planning.plan_window_set(80000, 32000, 1, 1)
    assert d[0][0] == 5 and d[0][2:4] == [ws0.m_min, ws0.count] and d[0][4] == -1
    assert d[1][0] == 6 and d[1][2:4] == [ws1.m_min, ws1.count] and d[1][4] == 9 and d[1][5] == d[0][1]   # distractor = key of unit 0
    assert d[2][0] == -1 and d[2][4] == -1
    assert lines[-1] == "hits 2 misses 2 resident 2"


def test_sim_state_columns_from_a_c_program(tmp_path):
    """ss_ctx_sims_units from C: azimuth = -rot mod 360 picks the row inside the pair's group of 4, the multi-second
    clip's window follows and advances _audio_index (simulator.py:629-635), an env past its duration is silent, and a
    pair that is not resident is reported without advancing anything."""
    import subprocess
    exe = _build_client(tmp_path, with_hip=False)
    out = subprocess.run([str(exe), "sims"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "step 0 misses 0"
    assert lines[1] == "env 0: sound 0 t0 0 rir 40 audio_index 0"              # 1-s clip: window 0, index untouched
    assert lines[2] == "env 1: sound 1 t0 16000 rir 11 audio_index 2"          # rot 90 -> azimuth 270 -> row 8 + 3; 1 -> 2 of 5
    assert lines[3] == "env 2: sound 0 t0 0 rir -1 audio_index 4"              # step 9 > duration 5: silent
    assert lines[4] == "step 1 misses 1"                                       # env 2's new pair (0, 0) is not in the table
    assert lines[6] == "env 1: sound 1 t0 32000 rir 11 audio_index 2"          # window 2 now; nothing advanced on a miss


@pytest.mark.gpu
def test_context_api_from_a_c_program_on_gpu(tmp_path):
    """GPU half: the same C client uploads a bank with the HIP C API, calls ss_ctx_observe once and its outputs match the
    reference-run vectors / the oracle."""
    import subprocess
    import numpy as np
    from golden_util import case_inputs, case_outputs
    from oracle import ss_oracle as O
    exe = _build_client(tmp_path, with_hip=True)
    d = case_inputs("clip1s")
    m = case_inputs("multi_L1.0_i2")
    sr = d["sr"]
    srcs = [d["source"], m["source"]]
    rirs = [d["rir"], m["rir"]]
    cap = sr
    bank = np.zeros((2, 2, cap), np.float32)
    for i, r in enumerate(rirs):
        bank[i, :, :r.shape[0]] = r.T
    sound, t0, ridx = np.array([0, 1, 0], np.int32), np.array([0, 2 * sr, 0], np.int32), np.array([0, 1, -1], np.int32)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([sr, 2], np.int32).tobytes())
        f.write(np.array([len(s) for s in srcs], np.int32).tobytes())
        for s in srcs:
            f.write(np.ascontiguousarray(s, np.float32).tobytes())
        f.write(np.array([2, cap], np.int32).tobytes())
        f.write(np.array([r.shape[0] for r in rirs], np.int32).tobytes())
        f.write(bank.tobytes())
        f.write(np.array([3], np.int32).tobytes())
        f.write(sound.tobytes()); f.write(t0.tobytes()); f.write(ridx.tobytes())
    out = subprocess.run([str(exe), "observe", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    raw = np.fromfile(tmp_path / "out.bin", np.float32)
    ag = raw[:3 * 2 * sr].reshape(3, 2, sr)
    sg = raw[3 * 2 * sr:].reshape(3, 65, 26, 2)
    for n, name in enumerate(("clip1s", "multi_L1.0_i2")):
        ref_a, ref_s, stride = case_outputs(name)
        assert O.relerr(ag[n][:, ::stride], ref_a) <= 1e-4 and O.relerr(sg[n], ref_s) <= 1e-4
    assert not ag[2].any() and not sg[2].any()


@pytest.mark.gpu
def test_integration_md_context_stub_runs_as_written():
    """The reference-side ctypes binding printed in INTEGRATION.md section 2a, executed verbatim (extracted from the
    document) against the built library, checked against the oracle."""
    import numpy as np
    import torch
    from oracle import ss_oracle as O
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("# soundspaces/_ss_hip.py  (reference-side binding)"):]
    block = block[:block.index("```")]
    block = block.replace('ctypes.CDLL("libss_hip.so")', f'ctypes.CDLL("{_lib.SO_PATH}")')
    sr = 16000
    rng = np.random.default_rng(0)
    clips = {f"s{i}": c for i, c in enumerate(O.synth_sources(rng, sr, k=2))}
    rirs = O.synth_rir(rng, sr, n=3)                                   # planar [3, 2, sr]
    bank = torch.from_numpy(rirs).to("cuda:0")
    ns = {"sr": sr, "source_sound_dict": clips, "bank": bank, "cap": sr,
          "bank_len": torch.full((3,), sr, dtype=torch.int32, device="cuda:0")}
    exec(block, ns)
    sg = torch.empty((3, 65, 26, 2), device="cuda:0")
    ns["observe"]([0, 1, 0], [0, 0, 0], [2, 0, -1], sg)
    torch.cuda.synchronize()
    sg = sg.cpu().numpy()
    for n, (s, h) in enumerate(((0, 2), (1, 0))):
        a = O.compute_audiogoal(clips[f"s{s}"], np.ascontiguousarray(rirs[h].T), sr)
        assert O.relerr(sg[n], O.compute_spectrogram(a)) <= 1e-4
    assert not sg[2].any()
