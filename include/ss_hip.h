/* ss_hip.h — C ABI of libss_hip.so: the MI355X (gfx950) audio-observation path of SoundSpaces.
 *
 * Every entry point replaces one piece of reference Python (paths relative to the
 * facebookresearch/sound-spaces checkout) and is what a ctypes/cffi binding on the
 * reference side would bind:
 *
 *   ss_fftconv_binaural_f32  <- SoundSpacesSim._compute_audiogoal: the two
 *                               scipy.signal.fftconvolve calls + slicing, all three windowing
 *                               branches, the distractor add and the silent / empty-RIR zeros
 *                               (soundspaces/simulator.py:608-666); also
 *                               ContinuousSoundSpacesSim._convolve_with_rir
 *                               (soundspaces/continuous_simulator.py:428-456)
 *   ss_spectrogram_f32       <- SpectrogramSensor.compute_spectrogram
 *                               (soundspaces/tasks/nav.py:86-100): librosa.stft(512,160,400) ->
 *                               abs -> block_reduce(4,4,mean) -> log1p -> stack(axis=-1)
 *   ss_audio_obs_f32         <- get_current_spectrogram_observation on a cache miss
 *                               (soundspaces/simulator.py:690-701), both stages fused
 *   ss_intensity_f32         <- Intensity.get_observation (ss_baselines/av_wan/avwan_sensors.py:91-100)
 *   ss_gccphat_f32           <- extension, not in the reference (GCC-PHAT inter-aural feature of configs[4])
 *   ss_logmel_f32            <- extension, not in the reference (log-mel front end named by the north star)
 *   ss_source_windows_f32    <- the FFT of the source clip that fftconvolve recomputes on every
 *                               call (simulator.py:630) hoisted out and cached per (sound, window)
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller (no torch types, no host
 * buffers); `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 * asynchronous on that stream; return 0 on success, SS_EINVAL for bad arguments, or the negated
 * hipError_t of the failing runtime call.  The library keeps only immutable per-device twiddle /
 * window tables, built on first use (thread-safe).
 */
#ifndef SS_HIP_H
#define SS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SS_EINVAL (-1)
#define SS_PAD_REFLECT 0   /* librosa < 0.10 default (the reference's era), torch.stft default */
#define SS_PAD_CONSTANT 1  /* librosa >= 0.10 default */
/* flags: promises from the caller that let the library pick a leaner kernel */
#define SS_FLAG_NO_DISTRACTOR 1  /* term 1 of every unit descriptor is absent ([4] == -1); it is then ignored */
/* SoundSpaces 2.0 CROSSFADE (soundspaces/continuous_simulator.py:47-53, 422-424): term 1 of a unit descriptor is the
 * PREVIOUS step's RIR (`_last_rir`), not a distractor: where it is present the row is
 *   out[:, n] = conv(term 1)[n] * (F - n)/F + conv(term 0)[n] * n/F   for n <= F = int(0.05 * out_len),
 *   out[:, n] = conv(term 0)[n]                                         beyond
 * (rows are 1 s long in both simulators, so out_len is the sampling rate); units whose term 1 is absent
 * (first step of an episode, `_last_rir is None`) are not blended.  One launch; with ss_audio_obs_f32 the spectrogram
 * is taken from the blended row.  F <= 2414 (sampling rates up to 48 kHz): the ramp waits in LDS while the row is convolved
 * with the current RIR; longer ramps are refused (SS_EINVAL). */
#define SS_FLAG_CROSSFADE 2
/* length-bucketed banks (ss_rir_bucket): every bank index the launch references lies in bucket 0, so the loop-free
 * kernels may serve it (the context's planner sets it per step) */
#define SS_FLAG_FIRST_BUCKET 4

/* Geometry constants of the partitioned convolution. */
int ss_block_len(void);        /* kB = 16384 real samples per partition block                     */
int ss_spec_floats(void);      /* floats per stored window spectrum = 2 * 16384                   */
int ss_version(void);

/* Build (idempotent) the constant tables for the current HIP device. */
int ss_init(void);
/* Rows longer than kB (44.1 kHz) make the library keep a scratch per (device, stream) for the block spectra of the rows in
 * flight (96 MiB per stream at 44.1 kHz without distractors), allocated by the stream's first such call (not inside a
 * hipGraph capture: warm the stream up first).  This frees all of them (after a device synchronise). */
int ss_release_scratch(void);

/* Source-window spectra.  win_desc[w] = {src_offset, src_len, start, wrap} (int32 x4):
 * window w holds samples src[src_offset + start + n], n in [0, 2*kB), zero outside [0, src_len)
 * (wrap != 0: indices >= src_len continue from the start of the clip, SS2.0 semantics).
 * For the convolution out[t] = sum_k h[k] x[t0+t-k] the window of partition offset m starts at
 * start = t0 + (m-1)*kB.  spec_out receives W * ss_spec_floats() floats (opaque kernel order). */
int ss_source_windows_f32(const float* src, const int* win_desc, float* spec_out, int n_windows,
                          void* stream);

/* Batched binaural convolution.  One unit = one (env, rotation) observation.
 * unit_desc[n] = 8 x int32, two terms k = 0 (source), 1 (distractor):
 *   [4k+0] RIR bank index, or -1 if the term is absent (both absent = silent unit -> exact zeros)
 *   [4k+1] slot (index into spec, in windows) of the spectrum for partition offset m_min
 *   [4k+2] m_min   [4k+3] number of consecutive offsets stored
 * RIR bank addressing: sample j of ear c of entry r is
 *   rir[r*rir_unit_stride + c*rir_chan_stride + j*rir_elem_stride], j < rir_len[r]
 *   (planar [R,2,L]: (2L, L, 1);  wav-interleaved [R,L,2]: (2L, 1, 2)).
 * Bank rows must be ZERO for rir_len[r] <= j < rir_cap (rir_cap = L of the bank): the kernel reads
 * up to rir_cap without per-sample length checks; rir_len only selects how many blocks are transformed.
 * out: [n_units, 2, out_len]; the first n_valid samples of every row are computed, the rest zeroed
 * (SS2.0 pads 0.25 s steps to 1 s).  n_valid <= 3*kB. */
int ss_fftconv_binaural_f32(const float* spec, const float* rir, const int* rir_len,
                            const int* unit_desc, float* out, int n_units,
                            long long rir_unit_stride, int rir_chan_stride, int rir_elem_stride,
                            int rir_cap, int n_valid, int out_len, int flags, void* stream);

/* Spectrogram of x [n_units, 2, len] -> out [n_units, 65, ceil((1+len/160)/4), 2] (channel-last). */
int ss_spectrogram_f32(const float* x, float* out, int n_units, int len, int pad_mode, void* stream);

/* Fused observation: convolution + spectrogram in ONE launch for rows of up to 3*kB samples (16 kHz: 1 block; 44.1 /
 * 48 kHz, the reference's Replica rate: 3 blocks, STFT streamed behind the output blocks).  audiogoal may be NULL: the
 * waveform then never leaves the CU - cross-faded rows (SS_FLAG_CROSSFADE) included, at either length.  Rows longer than kB
 * of which ONE block is rendered (n_valid <= kB: every SoundSpaces 2.0 step at 44.1 kHz) are served by the fused loop kernel
 * in one launch, audiogoal buffer or not (block spectra accumulated in registers; time-domain banks).  With an audiogoal
 * buffer, cross-faded rows with more than one rendered block take two launches (convolution, spectrogram), which is 20 %
 * faster than their one-launch form.  Pooled blocks that lie behind the rendered samples of a short step are exact zeros
 * and are written, not computed.  For rows longer than kB the library keeps a per-(device, stream) scratch for the block spectra of
 * the rows in flight (<= 2 x ceil(rir_cap/kB) x 128 KiB per CU), allocated on the stream's first such call.
 * Small steps of such rows (rows x output blocks <= the launch's share of the CUs: <= 42 units at 44.1 kHz) are rendered by one
 * workgroup per OUTPUT BLOCK (k_obs_blocks): the samples in front of a block boundary are handed to the next block's workgroup
 * through a per-(device, stream) area of 1.3 MB whose flags carry a launch counter - such a launch must not be replayed from a
 * captured hipGraph (the counter would repeat; the library itself never captures). */
int ss_audio_obs_f32(const float* spec, const float* rir, const int* rir_len, const int* unit_desc,
                     float* audiogoal, float* spectrogram, int n_units,
                     long long rir_unit_stride, int rir_chan_stride, int rir_elem_stride,
                     int rir_cap, int n_valid, int out_len, int pad_mode, int flags, void* stream);

/* ---- Spectral RIR bank ("store the RIR bank as half-spectra: x2 bytes, -50 % FLOPs") -----------------------
 * SoundSpaces 1.0 RIRs are static files (simulator.py:615-618), so the forward FFT of the RIR that fftconvolve repeats
 * on every step (:630) can be done ONCE when the bank is built.  ss_rir_spectra_f32 turns a planar time-domain bank
 * (sample j of ear c of entry r at rir[r*rir_unit_stride + c*rir_chan_stride + j], rows zero beyond their length up to
 * rir_cap) into hspec_out = n_entries * 2 * h_blocks * ss_spec_floats() floats, h_blocks = ceil(rir_cap / kB): the
 * block spectra of every (entry, ear) in the kernels' register order (opaque).  Synchronous (one-off, at bank load).
 * The *_spec_* entry points are ss_fftconv_binaural_f32 / ss_audio_obs_f32 reading that bank: same unit descriptors,
 * same rir_len (it still says how many blocks of an entry are non-zero), same results to fp32 rounding, no forward
 * FFT.  They read 2x the bytes per RIR; SS_FLAG_CROSSFADE is not supported (live SS2.0 RIRs have no static bank). */
int ss_rir_spectra_f32(const float* rir, float* hspec_out, int n_entries, long long rir_unit_stride,
                       int rir_chan_stride, int rir_cap, void* stream);
int ss_fftconv_binaural_spec_f32(const float* spec, const float* hspec, const int* rir_len, const int* unit_desc,
                                 float* out, int n_units, int h_blocks, int n_valid, int out_len, int flags,
                                 void* stream);
int ss_audio_obs_spec_f32(const float* spec, const float* hspec, const int* rir_len, const int* unit_desc,
                          float* audiogoal, float* spectrogram, int n_units, int h_blocks, int n_valid, int out_len,
                          int pad_mode, int flags, void* stream);

/* ---- Length-bucketed RIR bank ------------------------------------------------------------------------
 * The reference's RIRs are variable-length wav files (soundspaces/README.md:38-42, read at simulator.py:615-618; SS2.0's
 * ray-traced RIRs run to 4 s).  One capacity for every row means one long RIR reallocates the whole bank, multiplies the
 * HBM of every slot and pushes every launch off the loop-free kernel.  A bucketed bank keeps entries of similar length
 * together: bucket b holds bank indices [first, first + n_entries) as planar rows [n_entries, 2, cap] (zero beyond each
 * entry's length) in an allocation of its own, optionally with its spectral form (ss_rir_spectra_f32 of that bucket).
 * Buckets are ordered by `first` (bucket 0 starts at 0), ranges disjoint, at most 4; rir_len is ONE device array indexed by
 * the global bank index.  Unit descriptors are unchanged (they carry global indices).  Launches that promise
 * SS_FLAG_FIRST_BUCKET (and whose bucket 0 has cap <= kB) run the loop-free kernel. */
typedef struct ss_rir_bucket {
    const float* rir;     /* device, [n_entries, 2, cap] */
    const float* hspec;   /* device, [n_entries, 2, ceil(cap/kB), ss_spec_floats()] or NULL */
    int first;            /* global bank index of entry 0 of the bucket */
    int n_entries;
    int cap;              /* samples per (entry, ear) row, even */
    int reserved;
} ss_rir_bucket;
/* `buckets` is a HOST array of n_buckets descriptors.  The spectral kernels are used when every bucket carries hspec and
 * the launch is not cross-faded. */
int ss_fftconv_binaural_buckets_f32(const float* spec, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len,
                                    const int* unit_desc, float* out, int n_units, int n_valid, int out_len, int flags,
                                    void* stream);
int ss_audio_obs_buckets_f32(const float* spec, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len,
                             const int* unit_desc, float* audiogoal, float* spectrogram, int n_units, int n_valid,
                             int out_len, int pad_mode, int flags, void* stream);

/* EXTENSION, one pass for every STFT-derived feature (BASELINE.json configs[4] "GCC-PHAT + log-mel fused sensor"): each
 * (unit, ear, frame) of x [n_units, 2, len] is framed, windowed and transformed ONCE (both ears in one wave), and from that
 * one spectrum the kernel writes any subset of
 *   spectrogram [n_units, 65, T4, 2]            as ss_spectrogram_f32 (the reference's compute_spectrogram, nav.py:86-100)
 *   logmel      [n_units, n_mels, T, 2]         as ss_logmel_f32  (n_mels <= 64; larger banks: ss_logmel_f32)
 *   gccphat     [n_units, 2*max_lag+1, T]       as ss_gccphat_f32
 * (NULL = not wanted; at least one).  Same arguments and results as the three stand-alone entry points, which each re-read
 * the waveform and redo the STFT. */
int ss_audio_features_f32(const float* x, int n_units, int len, int pad_mode, float* spectrogram, float* logmel,
                          const int* mel_start, const float* mel_w, int n_mels, int max_len, float mel_eps,
                          float* gccphat, int max_lag, float gcc_eps, void* stream);

/* ---- The loop-free case on the 512-thread FFT core (round 4; csrc/ss_fft_core32.hpp) ------------------------------------
 * Rows of ONE partition block from a time-domain bank of rir_cap <= kB, no distractor / cross-fade terms - SoundSpaces 1.0
 * with 1-s clips (simulator.py:629-632) - on a 512-thread / 32-values-per-thread transform (16384 = 32*32*16: two LDS
 * exchanges per transform instead of three, half the workgroup barriers).  Window spectra for it come from
 * ss_source_windows32_f32 (same descriptors and size as ss_source_windows_f32, that core's own register order: the two
 * formats are not interchangeable).  audiogoal or spectrogram may be NULL (not both).  Results equal ss_audio_obs_f32's to
 * fp32 rounding.  Measured A/B against the 1024-thread kernels: profiles/r4/NOTES.md section 1. */
int ss_source_windows32_f32(const float* src, const int* win_desc, float* spec_out, int n_windows, void* stream);
int ss_audio_obs32_f32(const float* spec32, const float* rir, const int* rir_len, const int* unit_desc, float* audiogoal,
                       float* spectrogram, int n_units, long long rir_unit_stride, int rir_chan_stride, int rir_elem_stride,
                       int rir_cap, int n_valid, int out_len, int pad_mode, void* stream);

/* av_wan Intensity sensor (ss_baselines/av_wan/avwan_sensors.py:91-100) on audiogoal [n_units, 2, len]:
 * onset = min over ears of the first sample > 0.1*max, out[n] = mean(x[:, onset:onset+num_frame]**2). */
int ss_intensity_f32(const float* audiogoal, float* out, int n_units, int len, int num_frame, void* stream);

/* EXTENSION (no counterpart in the reference; BASELINE.json north_star "log-mel"): log-mel spectrogram of
 * x [n_units, 2, len] -> out [n_units, n_mels, 1 + len/160, 2] (channel-last, no pooling):
 *   out = log( sum_k W[j][k] * |STFT(x)[k]|^2 + eps ), STFT framing exactly as ss_spectrogram_f32.
 * The filter bank is band-sparse: band j covers bins mel_start[j] .. mel_start[j] + max_len - 1 with weights
 * mel_w[j*max_len + i] (zero padded).  n_mels <= 128; max_len a multiple of 4, <= 64 (32 bands at 48 kHz; 20-band banks are wider); n_mels*max_len <= 4096;
 * mel_start[j] a multiple of 4 in [0, 256] (pad the band with leading zero weights; the kernel reads 16 bytes at a
 * time and sees zeros beyond bin 256); mel_w 16-byte aligned.  mel_start's range is the caller's responsibility. */
int ss_logmel_f32(const float* x, float* out, int n_units, int len, int pad_mode, const int* mel_start,
                  const float* mel_w, int n_mels, int max_len, float eps, void* stream);

/* EXTENSION (no counterpart in the reference; BASELINE.json configs[4] "GCC-PHAT"): generalised cross-correlation
 * with phase transform between the two ears, per STFT frame (framing exactly as ss_spectrogram_f32):
 *   G[k] = X_left[k] conj(X_right[k]);  g = irfft(G / (|G| + eps), 512);  out[n][i][t] = g[(i - max_lag) mod 512]
 * x [n_units, 2, len] -> out [n_units, 2*max_lag + 1, 1 + len/160].  1 <= max_lag <= 32, eps > 0. */
int ss_gccphat_f32(const float* x, float* out, int n_units, int len, int pad_mode, int max_lag, float eps,
                   void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Context API: the self-contained entry.  One call per vector step does what the reference does per env
 * in Python -- SoundSpacesSim._compute_audiogoal (soundspaces/simulator.py:608-666) /
 * ContinuousSoundSpacesSim._compute_audiogoal + _convolve_with_rir (soundspaces/continuous_simulator.py:413-456) and
 * SpectrogramSensor.compute_spectrogram (soundspaces/tasks/nav.py:86-100) -- for ALL envs: the caller hands over what a
 * simulator knows (which sound, which clip window, which RIR), the library plans the partitions, keeps the
 * source-window spectra it needs in a bounded cache, uploads the descriptors and launches.  A C caller therefore needs
 * neither ss_amd/planning.py nor the opaque spectrum slots of the entry points above.
 *
 * A context is bound to the HIP device that is current when it is created.  Calls must come from one host thread at a time;
 * consecutive calls may name different streams (the library orders the shared window-spectra pool between them: on a
 * stream change it records an event on the PREVIOUS call's stream, so a stream handed to ss_ctx_observe must stay alive
 * until a later call on another stream has returned, or the context is destroyed), and ss_ctx_set_overlap lets the library
 * alternate two internal streams itself.  hipStreamPerThread is not accepted for rows longer than one block (one handle,
 * a different stream per thread).  ss_ctx_plan is a planner-only / test entry: after it the next ss_ctx_observe synchronises
 * the device and starts from an empty window cache.
 * ---------------------------------------------------------------------------------------------------------------*/
typedef struct ss_ctx ss_ctx;

/* The units of one step, struct-of-arrays, HOST memory, n entries per column; optional columns may be NULL.
 * unit i:  out[c,t] = sum_k rir[c,k] * x[t0 + t - k], t < n_valid (rest of the 1-s row zero), x = clip `sound`,
 *   x[<0] = 0 and past the clip end 0 or -- contexts created with wrap_mode 1, units with wrap != 0 -- the clip again
 *   from its start (the reference wraps only in its steady branch, continuous_simulator.py:438-447).
 *   t0 per reference branch: 0 (1-s clip, simulator.py:629-632); _audio_index * sr (multi-second, :634-647);
 *   _current_sample_index (continuous_simulator.py:432).
 *   rir < 0: silent unit (simulator.py:610-612) -> exact zeros.
 *   dis_sound / dis_rir: distractor (whole clip through a second RIR, added; simulator.py:649-664), -1 = none.
 *   last_rir: the previous step's RIR for SS2.0 CROSSFADE (see SS_FLAG_CROSSFADE), -1 = none; last_wrap = the branch
 *   flag of THAT RIR (defaults to wrap).  A step has either distractors or cross-fades, not both. */
typedef struct ss_units {
    const int* sound;
    const int* t0;
    const int* rir;
    const int* dis_sound;            /* optional */
    const int* dis_rir;              /* optional */
    const int* last_rir;             /* optional */
    const unsigned char* wrap;       /* optional; NULL = 1 for every unit */
    const unsigned char* last_wrap;  /* optional; NULL = same as wrap */
} ss_units;

/* n_valid = samples computed per row: sampling_rate (SoundSpacesSim) or int(sampling_rate * STEP_TIME) (Continuous);
 * rows are sampling_rate long.  wrap_mode: 0 = SoundSpaces 1.0, 1 = SoundSpaces 2.0 semantics (see ss_units).
 * max_window_sets = initial capacity of the window-spectra cache in (sound, t0) keys (128 KiB of HBM per stored
 * window; <= 0: 256); the cache is LRU and only grows when one step itself needs more keys than it holds. */
int ss_ctx_create(ss_ctx** ctx, int sampling_rate, int n_valid, int pad_mode, int wrap_mode, int max_window_sets);
int ss_ctx_destroy(ss_ctx* ctx);
/* Register a mono clip, float32, already at the simulator rate (the reference's _source_sound_dict,
 * simulator.py:595-600).  `clip` is host memory (on_device = 0) or device memory.  Returns the sound id >= 0. */
int ss_ctx_add_source(ss_ctx* ctx, const float* clip, int len, int on_device);
int ss_ctx_add_source_len(ss_ctx* ctx, int len);   /* planner-only registration (no device memory; for ss_ctx_plan) */
/* RIR bank in device memory, addressing as for ss_fftconv_binaural_f32; the pointers are borrowed. */
int ss_ctx_set_rir_bank(ss_ctx* ctx, const float* rir, const int* rir_len, long long rir_unit_stride,
                        int rir_chan_stride, int rir_elem_stride, int rir_cap);
/* Optional spectral form (ss_rir_spectra_f32) of the bank given to ss_ctx_set_rir_bank: steps without a cross-fade then
 * run the *_spec_* kernels.  hspec = NULL: back to the time-domain kernels.  Borrowed pointer. */
int ss_ctx_set_rir_spectra(ss_ctx* ctx, const float* hspec, int h_blocks);
/* The bank as length buckets (see ss_rir_bucket; replaces the two calls above for such banks; the descriptor array is copied,
 * the device pointers are borrowed).  Call again whenever a bucket is (re)allocated. */
int ss_ctx_set_rir_buckets(ss_ctx* ctx, const ss_rir_bucket* buckets, int n_buckets, const int* rir_len);
/* One step.  audiogoal [n,2,sr] and spectrogram [n,65,T4,2] are device buffers; either may be NULL (not both).
 * Asynchronous on `stream`; the host arrays of `units` may be reused as soon as the call returns. */
int ss_ctx_observe(ss_ctx* ctx, const ss_units* units, int n, float* audiogoal, float* spectrogram, void* stream);
/* One step plus its STFT-derived extension features (BASELINE.json configs[4]: savi, "GCC-PHAT + log-mel fused sensor"): as
 * ss_ctx_observe, then ss_audio_features_f32 over the step's waveform on the SAME stream (in overlap mode: the same internal
 * lane), so the features need no join of their own.  audiogoal must be given (the features read it); logmel / gccphat of
 * `f` may each be NULL.  When a spectrogram is asked for as well it is pooled by the feature kernel (which holds every
 * frame's spectrum of both ears anyway) and the convolution launch runs without its fused STFT phase: 96.5 instead of 108 us
 * per 256-env savi step. */
typedef struct ss_features {
    float* logmel;            /* [n, n_mels, 1 + sr/160, 2] or NULL */
    const int* mel_start;     /* device, [n_mels]          (ss_logmel_f32's band-sparse filter bank) */
    const float* mel_w;       /* device, [n_mels, max_len] */
    int n_mels, max_len;
    float mel_eps;
    float* gccphat;           /* [n, 2*max_lag+1, 1 + sr/160] or NULL */
    int max_lag;
    float gcc_eps;
} ss_features;
int ss_ctx_observe_features(ss_ctx* ctx, const ss_units* units, int n, float* audiogoal, float* spectrogram,
                            const ss_features* f, void* stream);
/* Overlap mode.  ss_ctx_set_overlap(ctx, n), n = 2 .. 4: consecutive ss_ctx_observe calls run on n internal streams in turn
 * (2: the head of step k+1 under the tail of step k; 3 - 4: for steps of few rows, where several launches fit the chip side by
 * side - fused rows are then split over fewer workgroups, ConvParams::parts_log2 - and the caller has that many to issue), each
 * ordered behind what the CALLER's stream holds at the time of the call (the consumers of the output rows it overwrites,
 * uploads of the RIR rows it reads), so the head of step k+1 (descriptor and row loads: HBM latency, nothing to compute)
 * overlaps the tail of step k (STFT: no memory traffic) - what the reference's serial per-env loop
 * (ss_baselines/common/sync_vector_env.py:397-410) cannot do.  Results become visible to a stream through
 * ss_ctx_join(ctx, stream) (the stream waits for every step issued so far); a caller that needs step k before it issues
 * step k+1 joins every step and gets the single-stream behaviour (joins from several streams each wait).  n_streams = 1
 * switches back (synchronises the device).
 * Steps must write disjoint output rows while they are in flight (rollout rows are), and a caller that REWRITES bank rows
 * (eviction: a new pose's RIR over an old entry) while steps are in flight joins first on the stream that carries the
 * rewrite - a step still running on a lane may read the old entry.  The library's own loaders (ss_ctx_observe_requests_load,
 * ss_ctx_load_rir_files) order their scatter behind every step issued so far themselves.
 * Threading: the ordering behind the caller's stream is elided when that stream is IDLE at the time of the call (one
 * hipStreamQuery instead of an event record + a stream wait).  That test is only sound for a single-threaded user of `stream`:
 * a second host thread that enqueues work on the same stream between the query and the lane's launch is NOT ordered in front
 * of the step.  Callers that share the stream between threads must serialise ss_ctx_observe with those enqueues (the
 * reference has one env loop per process: ss_baselines/common/sync_vector_env.py:397-410).  A stream that is being captured
 * into a graph is never queried (hipStreamIsCapturing first); the step is then always fenced. */
int ss_ctx_set_overlap(ss_ctx* ctx, int n_streams);
int ss_ctx_join(ss_ctx* ctx, void* stream);
/* A context that holds BOTH bank forms (ss_ctx_set_rir_bank + ss_ctx_set_rir_spectra) renders from the spectral one.  For
 * rows of one block (16 kHz) that pays for SMALL steps only - no forward FFT on a chip that is mostly idle: 12.9-15.6 us
 * against 17.1-19.0 us per step of 1-32 envs, the reference's 5-10 envs per GPU (ss_baselines/av_nav/config/audionav/
 * {replica,mp3d}/train_telephone/audiogoal_depth_ddppo.yaml:3) - while at 128 envs the forward FFT hides under the rows' loads
 * and the spectral rows are twice the bytes.  max_units > 0: steps of more than max_units units of one-block rows take the
 * time-domain rows - unless the step has distractor terms (two forward transforms per row do not hide under one row's load:
 * savi's 256-env step is 87.9 against 96.3 us); rows of several blocks (44.1 / 48 kHz) are not affected.  0 (default): the
 * spectral form whenever set. */
int ss_ctx_set_spectral_policy(ss_ctx* ctx, int max_units);
/* Small steps are rendered by several workgroups per row while CUs would idle (ConvParams::parts_log2).  A caller that keeps
 * SEVERAL launch sources busy at once - e.g. two env groups stepped alternately, each with a context of its own on its own
 * stream (the double-buffered sampler of bench.py's `dependent.two_groups`; ss_baselines/common/sync_vector_env.py has one
 * group) - says so here: every launch of this context then splits its rows over 1 / n_sources of the chip, so that the
 * sources' launches run side by side instead of queueing behind each other.  0 (default): the context's own lane count. */
int ss_ctx_set_chip_share(ss_ctx* ctx, int n_sources);
/* One step straight from the simulators' state, struct-of-arrays (ss_amd/vector.py::VectorSimState: the int64 columns a
 * vector env keeps per env; HOST memory, n entries each).  Does, for all envs at once, what SoundSpacesSim does per env:
 *   silent = step_count > duration (simulator.py:610) or sound < 0;   t0 = clip is 1 s ? 0 : audio_index * sr (:629-634);
 *   audio_index = (audio_index + 1) % (clip_len / sr) for multi-second clips that are not silent (:635; written back);
 *   azimuth = -rot mod 360 (:573);   RIR slot = index[scene][recv][src] + azimuth / (360 / azimuths)  (the pair's azimuths
 *   sit in adjacent bank rows; table entry -1 = pair not resident);   distractor likewise from dis_sound / dis_src.
 * If some non-silent env's pair is not resident nothing is launched and nothing advanced: their env indices go to
 * miss_out (capacity n), *n_miss > 0 and the call returns 0 - load the pairs, fix the table, call again. */
typedef struct ss_sim_columns {
    const long long* sound;          /* sound id per env, -1 = unknown                     */
    long long* audio_index;          /* in / out                                           */
    const long long* step_count;
    const long long* duration;
    const long long* recv;
    const long long* src;
    const long long* rot;            /* _rotation_angle, degrees                           */
    const long long* scene;          /* scene id = index into index_off / index_dim        */
    const long long* dis_sound;      /* optional (HAS_DISTRACTOR_SOUND): both or neither   */
    const long long* dis_src;
    const int* index_flat;           /* concatenated [dim x dim] tables, first slot or -1  */
    const long long* index_off;      /* [n_scenes] offset of a scene's table in index_flat */
    const long long* index_dim;      /* [n_scenes] nodes per scene                         */
    int n_scenes;
    int azimuths;                    /* bank rows per (receiver, source) pair, e.g. 4      */
} ss_sim_columns;
int ss_ctx_observe_sims(ss_ctx* ctx, const ss_sim_columns* cols, int n, float* audiogoal, float* spectrogram,
                        int* miss_out, int* n_miss, void* stream);
/* The state -> unit columns step of ss_ctx_observe_sims alone (host only, needs no GPU; advances audio_index the same
 * way): units_out int32 [5, n] = sound, t0, rir, dis_sound, dis_rir. */
int ss_ctx_sims_units(ss_ctx* ctx, const ss_sim_columns* cols, int n, int* units_out, int* miss_out, int* n_miss);
/* One step from the packed REQUEST RECORDS of a multi-process vector env (ss_amd/deferred.py: every worker-side sensor
 * returns an AudioRequest whose `rec` is SS_REQ_WORDS int64 words; the trainer concatenates the N records of the step).  Does
 * in C++ what DeferredResolver._columns does in numpy: names travel as CRC-32 keys, ids and resident RIR files are found by
 * binary search in the caller's SORTED tables; reference: what _compute_audiogoal reads per env, simulator.py:608-666.
 *   words: [0] silent  [1] sound key  [2] t0  [3] RIR table key (= directory/azimuth)  [4] receiver  [5] source
 *          [6] distractor sound key or -1  [7] distractor source  [8] env  [9] reserved
 * A request that names an unknown sound / table, a pair that is not resident, or a row flagged stale is a MISS: their
 * indices go to miss_out (capacity n), *n_miss > 0, nothing is launched and the call returns 0 - register / load, call again. */
#define SS_REQ_WORDS 10
typedef struct ss_request_tables {
    const long long* sound_keys;   /* sorted */
    const long long* sound_ids;
    const long long* table_keys;   /* sorted */
    const long long* table_ids;
    const long long* pair_keys;    /* sorted: table id << 40 | receiver << 20 | source */
    const long long* pair_slots;   /* bank slot of the pair's RIR */
    const unsigned char* stale;    /* optional, [n_slots]: != 0 = the row must be reloaded before it is used */
    int n_sounds, n_tables, n_pairs, n_slots;
    /* optional: the caller's LRU clock.  last_used[slot] = tick is written for every bank slot (< n_slots) a request of the
     * step resolves to, so that a store evicting by recency still knows which rows the C path used (the lookups never pass
     * through the store).  NULL: nothing is written. */
    long long* last_used;
    long long tick;
} ss_request_tables;
int ss_ctx_observe_requests(ss_ctx* ctx, const long long* recs, int n, const ss_request_tables* tables, float* audiogoal,
                            float* spectrogram, int* miss_out, int* n_miss, void* stream);
/* Round 6: the same call with the miss path INSIDE it.  The reference reads a pose's RIR file on every cache-missing step
 * (wavfile.read at soundspaces/simulator.py:615-618); with ss_ctx_observe_requests a step that meets a pose whose file is not
 * resident costs three C calls and the store's Python bookkeeping between them (report -> ss_wav_read_rirs_f32 + scatter -> call
 * again).  Here the caller lends the library what that bookkeeping needs - the RIR directories, the sorted resident-pair arrays
 * with spare capacity, a stack of free bank entries, the bank and its length tables, a pinned staging block - and a step whose
 * only misses are poses that are not resident is served in ONE call: file names built, files read by the library's reader
 * straight into the staging block, one scatter launch into the free entries, the pair arrays extended in place, the step
 * launched.  What was loaded is reported (loaded_key / loaded_slot / loaded_frames, n_loaded) so that the caller's own tables
 * (key -> entry dictionaries, LRU order) follow; n_free and *n_pairs are updated.  Anything the fast path does not cover - an
 * unknown sound or directory, a stale row, no entry to be had (stack empty and no eviction arrays lent, or every occupied entry
 * in use by this very step), a file that is not a plain
 * float32 stereo wav / is missing / does not fit the rows, a launch that reads the spectral rows without `stage_desc` - changes NOTHING and is
 * reported exactly as ss_ctx_observe_requests reports it (miss_out, *n_miss > 0).  Arrays are HOST memory unless said
 * otherwise; `stage`, `stage_slot`, `stage_len` must be pinned (the scatter kernel reads them over the host link) and stay
 * untouched by the caller until the stream has run the launch (the library waits for its own previous use of them). */
typedef struct ss_miss_loader {
    const char* const* table_dirs;   /* [n_table_dirs]: directory of table id t ("<BINAURAL_RIR_DIR>/<scene>/<azimuth>")      */
    int n_table_dirs;
    long long* pair_keys;            /* the arrays ss_request_tables.pair_keys / pair_slots point at, writable, ...            */
    long long* pair_slots;
    int pair_cap;                    /* ... with room for pair_cap entries; ss_request_tables.n_pairs is updated                */
    int* free_slots;                 /* stack of free bank entries: the call pops free_slots[n_free - 1], ...                   */
    int n_free;                      /* in: entries on the stack; out: entries left                                             */
    float* bank;                     /* DEVICE: planar time-domain bank [entries][2][cap]                                        */
    long long bank_unit_stride;      /* floats between entries                                                                   */
    int bank_chan_stride, cap;
    int* dev_len;                    /* DEVICE: rir_len table of the bank                                                        */
    int* host_len;                   /* host mirror of it                                                                        */
    unsigned char* clipped;          /* [entries]: 1 = the stored row is shorter than its file (keep < frames)                  */
    unsigned char* spec_stale;       /* optional [entries]: 1 = the row's block spectra must be rebuilt before a spectral launch */
    int keep;                        /* frames kept per file (1-s clips only ever hear h[0:sr]) or -1 = whole files              */
    float* stage;                    /* PINNED: stage_rows rows of 2 * cap floats (wav layout)                                   */
    int* stage_slot;                 /* PINNED [stage_rows]                                                                      */
    int* stage_len;                  /* PINNED [stage_rows]                                                                      */
    int* stage_desc;                 /* optional, PINNED [stage_rows * 2 * ceil(cap / kB) * 5]: window descriptors of the new rows'
                                      * block spectra - with it, steps that read the SPECTRAL rows (ss_ctx_set_rir_spectra) are
                                      * served too: the new rows are transformed right behind the scatter                        */
    int stage_rows, threads;
    /* optional: eviction inside the call.  When the step needs more entries than the stack holds, the least recently used
     * occupied entries are reused - recency = ss_request_tables.last_used (entries stamped with the step's own tick are never
     * taken), ties by use_seq, oldest first: the policy of RirStore._take_slots.  Their pairs leave pair_keys / pair_slots in
     * the call; evicted_slot names them so that the caller drops its own records of the keys they held.  NULL: no eviction. */
    const unsigned char* used;       /* [n_slots of ss_request_tables]: 1 = the entry is occupied                                */
    const long long* use_seq;        /* [n_slots]: order of last use through the caller's own lookups                            */
    int* evicted_slot;               /* report, capacity evict_cap                                                               */
    int n_evicted, evict_cap;
    long long* loaded_key;           /* report, capacity loaded_cap: pair key, ...                                               */
    int* loaded_slot;                /* ... the entry it went to, ...                                                            */
    int* loaded_frames;              /* ... frames in its file (kept = min(frames, keep))                                        */
    int n_loaded, loaded_cap;
} ss_miss_loader;
int ss_ctx_observe_requests_load(ss_ctx* ctx, const long long* recs, int n, ss_request_tables* tables, ss_miss_loader* loader,
                                 float* audiogoal, float* spectrogram, int* miss_out, int* n_miss, void* stream);
/* The loader alone, for callers that name FILES (the eager call of an agent that has moved: simulator.py:615-618 reads the pose's
 * file on every cache-missing step): k wav files -> k bank entries of the lent store (free stack first, then the least recently
 * used occupied entries whose last_used is below `tick`; last_used / n_slots as in ss_request_tables, NULL = no eviction), one
 * scatter launch on `stream`, block spectra of the new rows when the context holds the spectral form; report in loaded_slot /
 * loaded_frames / evicted_slot.  pair_keys / pair_slots / table_dirs / loaded_key are not used.  Returns 0 = loaded, 1 = not for
 * this path (nothing changed: unusual file, no entry to be had, rows too short, the bank is not the context's), < 0 = error. */
int ss_ctx_load_rir_files(ss_ctx* ctx, ss_miss_loader* loader, const char* const* paths, int k, long long* last_used,
                          long long tick, int n_slots, void* stream);
/* The records -> unit columns step alone (host only, needs no GPU): units_out int32 [5, n] = sound, t0, rir, dis_sound, dis_rir. */
int ss_ctx_requests_units(ss_ctx* ctx, const long long* recs, int n, const ss_request_tables* tables, int* units_out,
                          int* miss_out, int* n_miss);
/* The planner alone (host only, needs no GPU): unit_desc_out int32 [n,8] as ss_fftconv_binaural_f32 takes them,
 * *flags_out the SS_FLAG_* of the launch, *n_new_windows_out the source windows whose spectra would be computed,
 * new_windows_out (optional, int32 [cap,5]) = {src_offset, src_len, start, wrap, pool slot} of those windows. */
int ss_ctx_plan(ss_ctx* ctx, const ss_units* units, int n, int* unit_desc_out, int* flags_out, int* n_new_windows_out,
                int* new_windows_out, int new_windows_cap);
/* out8 = {cache hits, misses, evictions, grows, capacity (keys), keys resident, pool slots per key, steps planned} */
int ss_ctx_stats(ss_ctx* ctx, long long* out8);

/* ---- RIR files -> staging rows (the step BEFORE the path: SURVEY 8(f)2) ---------------------------------------------
 * Replaces the reference's per-miss `scipy.io.wavfile.read(binaural_rir_file)` (soundspaces/simulator.py:615-618, float32
 * stereo files under <binaural_rir_dir>/<azimuth>/<recv>_<src>.wav) for bulk loads and the per-step pose misses of the
 * vector modes: n files are parsed (RIFF header, "fmt " / "data" chunks) and their first `keep` frames (keep < 0: all) are
 * read() straight into row i of `dst` on up to n_threads plain threads - HOST pointers here, typically a pinned staging
 * block that one H2D copy then moves.  Row i starts at dst + i * row_stride floats (row_stride >= 2 * cap) and is
 * wav-interleaved [cap][2] (planar = 0: the file's own layout, no transpose; the kernels read such rows with
 * rir_elem_stride = 2) or planar [2][cap] (planar = 1); rows are zero beyond the frames kept.
 * status_out[i]: 0 loaded; 1 not a plain little-endian float32 stereo RIFF/WAVE file (integer PCM, RIFX, malformed ...):
 * NOT interpreted here - route that file through the Python reader, which keeps scipy's semantics (ValueError -> zero RIR,
 * :619-621); 2 no frames (-> zero RIR, :622-624); 3 cannot be opened; 4 more frames to keep than `cap` (nothing read:
 * grow the rows and retry).  kept_out[i] = frames stored, frames_out[i] = frames in the file.  Rows with a non-zero
 * status are zero-filled.  Returns 0 / SS_EINVAL. */
int ss_wav_read_rirs_f32(const char* const* paths, int n, float* dst, long long row_stride, int cap, int keep, int planar,
                         int* kept_out, int* frames_out, int* status_out, int n_threads);
/* The miss path's last hop: n staged rows in wav layout - row i = frames j < lens[i] as (L, R) pairs at
 * staged[i*staged_row_stride + 2*j + c] - into the planar bank rows bank[slots[i]*unit_stride + c*chan_stride + j], zeros
 * behind each row's length up to `cap`, and lens[i] into bank_len[slots[i]] (bank_len may be NULL).  `staged`, `slots`
 * and `lens` may be PINNED HOST memory (the kernel pulls the samples over the host link: one launch, no staging copy) or
 * device memory (pageable host memory is refused: SS_EINVAL); bank / bank_len are device memory.  Asynchronous on `stream`: the caller keeps the staged block alive and
 * unchanged until the launch has run.  Replaces the reference's per-file host path simulator.py:615-624 -> numpy -> (no
 * device at all there) for what follows ss_wav_read_rirs_f32 / ss_rows_gather_f32.  Returns 0 / SS_EINVAL / -hipError_t. */
int ss_bank_scatter_rows_f32(const float* staged, long long staged_row_stride, const int* slots, const int* lens, int n,
                             float* bank, long long unit_stride, int chan_stride, int cap, int* bank_len, void* stream);
/* n HOST arrays -> n rows of a (pinned) staging block: row i = src[i][0 .. n_floats[i]) followed by zeros up to row_floats,
 * rows row_stride floats apart, on up to n_threads plain threads.  The live RIRs of a SoundSpaces 2.0 step (one new RIR per
 * env and step from the ray tracer, soundspaces/continuous_simulator.py:419) travel to the bank this way: one block, one
 * H2D copy.  Returns 0 / SS_EINVAL. */
int ss_rows_gather_f32(const float* const* src, const int* n_floats, int n, float* dst, long long row_stride, int row_floats,
                       int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* SS_HIP_H */
