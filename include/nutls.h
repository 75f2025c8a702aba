/*
 * nutls.h -- C ABI of the MI355X-native NUNet-TLS frame-by-frame forward pass.
 *
 * This is the drop-in boundary for ONE path of the reference: the per-frame model call
 *
 *     runner = interpreter.get_signature_runner('nutls_lstm_sm')          (interpreter_proposed.py:380)
 *     out    = runner(input=..., msfe6_ee_prev1=..., ..., msfe6_de_c=...)  (interpreter_proposed.py:215-350)
 *
 * (phone twin: Interpreter.runSignature(inputs, outputs, "nutls_lstm_sm"),
 *  mobile_app/.../RTSE_NUTLS_LSTM.java:571), i.e. the function
 * TFL_SIGNITURE.nutls_lstm of converter_proposed.py:188-867 evaluated by TF-Lite.
 *
 * Differences from the reference surface, by design:
 *   - B independent streams per handle (the reference is batch 1);
 *   - the 130 recurrent-state tensors (converter_proposed.py:27-186) stay resident in HBM
 *     between calls; the caller may read / write any of them by its signature name
 *     (the "cur -> prev" echo of interpreter_proposed.py:215-350 becomes a buffer flip);
 *   - weights come from a .nutlsw container (tools/convert_tflite_weights.py), not a .tflite.
 *
 * Plain pointers and sizes only.  Every entry returns 0 on success or a negative code;
 * nutls_last_error() returns a thread-local message for the last failure.
 * One handle = one GPU = one host thread at a time (like a TF-Lite Interpreter).
 */
#ifndef NUTLS_H_
#define NUTLS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nutls_handle nutls_handle;

enum {
  NUTLS_OK = 0,
  NUTLS_ERR_ARG = -1,      /* bad argument: unknown state name, size mismatch, null pointer (ValueError in TF-Lite) */
  NUTLS_ERR_WEIGHTS = -2,  /* malformed weight container or missing tensor */
  NUTLS_ERR_HIP = -3,      /* HIP runtime failure (message carries hipGetErrorString) */
  NUTLS_ERR_NO_DEVICE = -4 /* no usable gfx950 device: there is NO CPU fallback */
};

enum {
  NUTLS_VARIANT_LSTM = 0,     /* NUNet-TLS-LSTM, "proposed" (models/proposed.py; signature 'nutls_lstm_sm', 130 states) */
  NUTLS_VARIANT_BASELINE = 1  /* NUNet-TLS with the dilated-dense bottleneck (models/nunet_tls.py; signature 'nutls'
                               * of converter_nunet_tls.py:1542, 208 states).  No trained weights exist for it. */
};

#define NUTLS_BINS 256        /* network bins per frame (DC dropped, interpreter_proposed.py:212-213) */
#define NUTLS_LSTM_UNITS 21   /* models/proposed.py:21 */

/* Replaces: tf.lite.Interpreter(model_path) + allocate_tensors() (interpreter_proposed.py:374-375).
 * `weights`/`n_bytes`: the .nutlsw container bytes (copied; caller may free after return).
 * `batch`: number of independent streams B (>= 1).  `device`: HIP device ordinal. */
int nutls_create(const void* weights, size_t n_bytes, int variant, int batch, int device,
                 nutls_handle** out);

/* nutls_create with the fused kernel's plan chosen by the caller: streams_per_workgroup = 0 (the library's choice, = nutls_create), 1, 2 or
 * 4 (see nutls_streams_per_workgroup).  An explicit count the batch is not a multiple of, or one the variant has no plan for (3, 8, ...;
 * the baseline variant has the one-stream plan only) is NUTLS_ERR_ARG; an explicit count (1, 2, 4) with a container the fused kernel
 * cannot run -- conv kernels not stored as int8 -- is NUTLS_ERR_WEIGHTS with the reason in nutls_last_error (nutls_create would run such a
 * container on the per-layer kernels, mode 1).  An explicit request never silently becomes another plan or another kernel family. */
int nutls_create_plan(const void* weights, size_t n_bytes, int variant, int batch, int device, int streams_per_workgroup,
                      nutls_handle** out);

/* Replaces: del interpreter. */
int nutls_destroy(nutls_handle* h);

/* Replaces: one runner(...) call for all B streams (interpreter_proposed.py:215-350) with the
 * cur->prev echo folded in.  mag_in / mag_out: DEVICE pointers to [B, 256] float32
 * (row = stream; bins 1..256 of |STFT|).  Asynchronous on `stream` (a hipStream_t, may be NULL
 * for the default stream); state advances by one frame.  In the fused mode the kernel reads mag_in and writes mag_out
 * directly (no staging copy): both must stay valid, and mag_in unmodified, until the work queued on `stream` has run;
 * they may be the same buffer.  Side effect that differs by mode: modes 0-2 compute on the library's own [B,256] buffers
 * (nutls_io_buffers) and copy to / from the caller's, so those library buffers hold the frame afterwards; the fused mode
 * writes ONLY the mag_out it was given.  Whatever reads the library's mag_out buffer after a step (nutls_istft_hop) therefore
 * needs the step to have run on the library buffers (pass the pointers of nutls_io_buffers, which is what nutls_enhance_hop
 * does) -- after nutls_step(h, user_in, user_out) in the fused mode that buffer still holds the frame of the last step that
 * was given it. */
int nutls_step(nutls_handle* h, const float* mag_in, float* mag_out, void* stream);

/* Same with HOST buffers: H2D copy, step, D2H copy, synchronises before returning.  With pageable memory (malloc, numpy) the two copies go
 * through the runtime's staging buffers; with buffers from nutls_host_alloc they are plain DMA transfers, and in the fused mode there are no
 * copies at all: the kernel reads the frame from and writes the result to the pinned host buffers over the link (PCIe) itself. */
int nutls_step_host(nutls_handle* h, const float* mag_in, float* mag_out);

/* Page-locked, device-visible host memory for the buffers handed to nutls_step_host / nutls_enhance_hop_host / nutls_process_block_host
 * (what TF-Lite's interpreter.tensor(i) zero-copy view is to set_tensor / get_tensor in the reference's loop, interpreter_proposed.py:215-350:
 * the caller produces its frames in, and consumes its results from, memory the device can reach).  NULL on failure (nutls_last_error).
 * A handle is not needed; the memory is visible to every device. */
void* nutls_host_alloc(size_t bytes);
void nutls_host_free(void* p);

/* ---- STFT front end / inverse-STFT back end on the device -----------------
 * Replaces the numpy part of the reference's real_time_speech_enhancer loop
 * (dnn_model/interpreter_proposed.py:203-213 analysis, :352-365 synthesis): frame 512, hop 256,
 * analysis window = periodic Hann with end taps 1e-7 (:20-22), inverse window (:24-26), overlap-add.
 * The library keeps, per stream, the previous hop and the overlap tail (zeroed by nutls_reset).
 *
 * nutls_enhance_hop: pcm_in / pcm_out are DEVICE pointers to [B, 256] float32 (row = stream, one hop of
 * 256 samples): analysis -> model step -> synthesis, asynchronous on `stream`.  Like the reference loop the
 * output lags the input by one hop (the first returned hop is the leading half-window).
 * dc_mode: how bin 0 is re-created behind the 256-bin model. */
#define NUTLS_DC_EDGE 0   /* PC loop: np.pad(..., mode='edge'), interpreter_proposed.py:352-353 */
#define NUTLS_DC_ZERO 1   /* phone: zero, mobile_app/.../RTSE_NUTLS_LSTM.java:677 */
int nutls_enhance_hop(nutls_handle* h, const float* pcm_in, float* pcm_out, int dc_mode, void* stream);
/* Same with HOST buffers (H2D, pipeline, D2H, synchronises). */
int nutls_enhance_hop_host(nutls_handle* h, const float* pcm_in, float* pcm_out, int dc_mode);
/* The two halves on their own (testing, custom models): analysis of one hop into the library's mag_in buffer
 * (+ phase kept inside), synthesis of one hop from the library's mag_out buffer.  Device pointers. */
int nutls_stft_hop(nutls_handle* h, const float* pcm_in, void* stream);
int nutls_istft_hop(nutls_handle* h, float* pcm_out, int dc_mode, void* stream);

/* ---- Offline / block mode ------------------------------------------------
 * One utterance, up to `max_frames` consecutive frames per call: the frame index takes the place of the stream
 * index (a conv layer's previous-frame tap is the same tensor one frame earlier), every conv-like layer runs once
 * per block over all frames, only the 13 LSTM recurrences are scanned sequentially.  Same function as feeding the
 * frames one by one to a batch-1 streaming handle (the offline formulation of models/proposed.py:284-625 with the
 * streaming CTFA of converter_proposed.py:258-262); the state carries over from block to block inside the handle
 * and is zeroed by nutls_reset(h, -1).  mag_in / mag_out: DEVICE pointers to [n_frames, 256] float32. */
int nutls_create_offline(const void* weights, size_t n_bytes, int max_frames, int device, nutls_handle** out);
/* The same with a BATCH dimension -- the offline forward of the reference takes [B, T, ...] (models/proposed.py:284-625): `utterances`
 * (1 .. 256, utterances x max_frames <= 65536) independent utterances per handle, every call processes the same number of frames of each:
 * mag_in / mag_out are [utterances, n_frames, 256].  Every conv-like layer is ONE launch over the frames of all utterances, the 13 LSTM
 * recurrences of the utterances are scanned side by side (one wavefront each).  Each utterance carries its own state from block to block;
 * nutls_reset(h, u) zeroes utterance u (-1: all), nutls_state_get / _set take [utterances, ...] buffers, nutls_state_get_all(h, u, ...) one
 * utterance's.  Results per utterance equal those of a one-utterance handle (same kernels, same tilings). */
int nutls_create_offline_batch(const void* weights, size_t n_bytes, int max_frames, int utterances, int device, nutls_handle** out);
int nutls_process_block(nutls_handle* h, const float* mag_in, float* mag_out, int n_frames, void* stream);
/* The frequency-attention branch of CTFA in offline handles:
 *   NUTLS_CTFA_FRAME     (default) what the frame-wise graph computes: the branch sees TA/32 (ctfa_rt with T = 1,
 *                        proposed.py:162-196) -- offline results equal the streaming ones;
 *   NUTLS_CTFA_CAUSAL32  the offline / training model (`ctfa`, proposed.py:125-160, :143-147): the mean of the time
 *                        attention over the last 32 real frames (zeros before the utterance starts).  The 31-frame
 *                        history carries from block to block; switching modes or nutls_reset clears it. */
#define NUTLS_CTFA_FRAME 0
#define NUTLS_CTFA_CAUSAL32 1
int nutls_offline_set_ctfa_mode(nutls_handle* h, int mode);
/* The same choice for any handle.  Offline handles: as above.  Streaming handles (fused kernel, mode 3, only): NUTLS_CTFA_CAUSAL32 keeps
 * the time attention of the last 32 frames of every stream and stage in the library (not a signature tensor: the reference's streaming
 * graph has no such state) and feeds the frequency branch their mean -- streaming then equals the offline / training
 * model's attention (proposed.py:143-147) frame by frame.  nutls_reset clears a stream's history, a mode switch everyone's. */
int nutls_set_ctfa_mode(nutls_handle* h, int mode);
/* Block pipeline of an offline handle.  Only the 13 LSTM recurrences are serial over the frames of a block, and every
 * layer is causal in time, so a block is cut into `chunks` runs of consecutive frames: chunk 0 executes on the caller's
 * stream, chunk c > 0 on its own HIP stream one bottleneck behind chunk c-1 (it needs that chunk's last frame:
 * previous-frame taps, h / c, time-attention history), and the scans of one chunk overlap the convolutions of the others.
 * Results do not depend on the chunk count.  chunks: 1 .. 16, or 0 (default) = chosen from the block length (2 from 256
 * frames on, 3 from 768: the GPU serves four compute queues at a time, a pipeline with more queues than that
 * serialises -- 4 chunks need GPU_MAX_HW_QUEUES >= 8 and are no faster than 3). */
int nutls_offline_set_pipeline(nutls_handle* h, int chunks);
/* Same with HOST buffers (synchronises). */
int nutls_process_block_host(nutls_handle* h, const float* mag_in, float* mag_out, int n_frames);

/* Library-owned device staging buffers [B,256]; stepping on them avoids the D2D copies and lets
 * the captured hipGraph run with no per-call parameter update. */
int nutls_io_buffers(nutls_handle* h, float** mag_in, float** mag_out);

/* Execution mode of nutls_step:
 *   3            fused kernel: ONE launch per frame, one 512-thread workgroup per stream (per two / four streams where the library's
 *                cost model finds a packed plan faster -- more streams than CUs: nutls_streams_per_workgroup); every op of the step
 *                is its own specialised instruction stream (static schedule, no plan decoding).  Both variants
 *                have one; it keeps the conv kernels int8 on the device, so it exists for handles made from a
 *                container with int8 conv kernels (what the reference's .tflite stores) and is their default;
 *   1            one kernel per layer, the ~160 launches captured in a hipGraph (one per state parity);
 *   0            one kernel per layer, plain launches (handles made from containers without int8 conv kernels start in mode 1).
 * (2 was the plan-interpreter kernel of rounds 1-3; retired, nutls_set_mode(2) is an error.)  All of them compute the same function (tests/test_gpu_parity.py::test_execution_modes_agree). */
int nutls_set_mode(nutls_handle* h, int mode);
/* enable != 0: mode 1 (capture + replay); enable == 0: mode 0. */
int nutls_use_graph(nutls_handle* h, int enable);

/* State access by the reference's signature names.  `name` is any of the 130 state inputs
 * ("msfe6_ee_prev1" ... "msfe6_de_c", converter_proposed.py:27-186); the matching output spelling
 * ("..._cur1") is accepted as an alias.  Host buffer layout: [B, F, C] (conv states) or [B, 21],
 * float32, `n_floats` must equal B * per-stream size.  Synchronous.
 * (The fused kernel does not write the 40 conv-input states it never reads itself -- the echoes of the strided convs' inputs,
 * converter_proposed.py:226-231 -- on every frame: their rows exist a second time as skip-connection slices of other states, and the
 * library rebuilds them from there before any of the accessors below, a step of another mode or nutls_reset looks at the states.
 * What a caller sees is what the reference's runner returns, frame by frame; NUTLS_EAGER_STATES=1 makes every launch write everything.)
 * nutls_state_set and the causal32 CTFA of a streaming handle: the 31-frame time-attention history (nutls_set_ctfa_mode) is library state
 * outside the signature's tensors and nutls_state_set does not touch it -- the buffers are [B, ...], so "get, change stream b's row, set"
 * leaves the other B - 1 live streams exactly as they were.  To start a NEW utterance in stream b from loaded states, call
 * nutls_reset(h, b) first (zeroes b's states and its history), then set. */
int nutls_state_get(nutls_handle* h, const char* name, float* host_buf, size_t n_floats);
int nutls_state_set(nutls_handle* h, const char* name, const float* host_buf, size_t n_floats);

/* Every state tensor of ONE stream, concatenated in nutls_state_info order (n_floats = sum of the per-stream sizes):
 * what the signature runner hands back per frame, with a single device-to-host copy. */
int nutls_state_get_all(nutls_handle* h, int stream_idx, float* host_buf, size_t n_floats);

/* Enumerate the state tensors: count, then name + per-stream dims (F, C) or (21, 1). */
int nutls_state_count(nutls_handle* h);
int nutls_state_info(nutls_handle* h, int index, const char** name, int* dim0, int* dim1);

/* Zero the state of one stream (stream_idx >= 0) or of all streams (stream_idx < 0): the
 * all-zero seed of interpreter_proposed.py:36-198. */
int nutls_reset(nutls_handle* h, int stream_idx);

/* Copy the last step's value of an internal activation to the host (testing / debugging):
 * "<stage>.y" (CTFA output [B,F0,64]), "<stage>.up" (decoder up-sampled input [B,F0,128]),
 * "input_layer" ([B,256,64]).  Per-layer modes (0 / 1): the last stage's tensors ("msfe6_de.y", "msfe6_de.up": the layers share one scratch
 * tensor per kind) and "input_layer".  Fused mode: these tensors never leave LDS; with nutls_debug_trace(h, 1) every step of a one-stream
 * fused handle (<= 64 streams) runs on the library's profiling build of the step kernel, which copies them out -- "<stage>.y" of all 12
 * stages, "<stage>.up" of the 6 decoder stages, "input_layer" of the LAST step -- for layer-by-layer comparison with a reference trace
 * (tests/test_gpu_trace.py).  Testing / debugging only: the profiling build is a few percent slower. */
int nutls_debug_trace(nutls_handle* h, int enable);
/* Developer knobs of a handle (timing experiments; no effect on results).  "skew": start skew of the fused kernel's workgroups -- workgroup w
 * sleeps (w mod 4) * value * 64 clocks before its first op (0 = off, the default; NUTLS_FUSED_SKEW sets it at creation).  Experiment builds
 * of the step kernel (tools/exp/build_plan_lib.sh ... -DFZ_STOPAT=1) read the same field as "stop after this many ops", which is how
 * tools/exp/prod_timeline.py times variants of the UN-instrumented kernel op by op (the library's own kernel: nutls_profile_production).
 * Unknown names: NUTLS_ERR_ARG. */
int nutls_debug_knob(nutls_handle* h, const char* name, int value);
int nutls_debug_get(nutls_handle* h, const char* name, float* host_buf, size_t n_floats);

int nutls_batch(nutls_handle* h);
/* Streams one workgroup of the fused kernel steps (mode 3): 1, or -- LSTM variant, more streams than CUs -- 2 or 4: a packed plan
 * (csrc/fused_step_g2.hip / _g4.hip: the layers whose LDS images fit that often run the streams side by side on one position axis, sharing
 * the weight fetch and conversion; chosen by nutls_create so that rounds of workgroups x step time of the plan is smallest -- with 256 CUs:
 * 256 streams 1, 300 .. 512 2, 768 1, 1024, 1536 and 2048 2; the four-stream plan on request).  A stream's results do not depend on its slot or partners.
 * NUTLS_FUSED_STREAMS=1 at creation keeps the one-stream plan; nutls_create_plan chooses explicitly.  No reference counterpart (the reference steps one stream: interpreter_proposed.py:215). */
int nutls_streams_per_workgroup(nutls_handle* h);
int nutls_launches_per_step(nutls_handle* h);

/* Introspection of the per-frame launch plan (nutls_launches_per_step entries, in issue order):
 * layer name (reference Keras layer, e.g. "msfe6_en_spconv6"), kernel family (one family = one
 * kernel symbol), and the ALGORITHMIC work of that launch for the handle's batch:
 * flops = 2 * MACs, bytes = 4 * (input elements read incl. the previous-frame tap + output
 * elements written)  -- the "layer-fused" traffic model of SURVEY.md section 8(d). */
int nutls_launch_info(nutls_handle* h, int index, const char** layer, const char** family,
                      double* flops, double* bytes);

/* Run ONE step with plain launches on the library's own stream, bracketing every launch with
 * HIP events recorded on that stream; writes the milliseconds of each launch to ms[0..n).
 * Advances the state like nutls_step (input = the library's mag_in buffer).  Synchronous. */
int nutls_profile_step(nutls_handle* h, float* ms, int n);

/* Fused-mode twin: op count / names / algorithmic flops per stream of a variant's static schedule, and one profiled
 * step (workgroup 0 stamps every op boundary); microseconds per op to us[0..nutls_fused_num_ops(variant of h)). */
int nutls_fused_num_ops(int variant);
int nutls_fused_op_info(int variant, int index, const char** name, double* flops);
int nutls_profile_fused(nutls_handle* h, double* us, int n);
/* The same timeline from the UN-instrumented instruction stream (one-stream plan of the LSTM variant): the library's "stop twin" of the step
 * kernel -- the production code plus one scalar compare per op -- ends a launch in front of op N; cum_us[N], N = 0 .. nutls_fused_num_ops (= the
 * whole step), is the time per launch of `steps` back-to-back launches that end there (best of `reps` windows, HIP events), so
 * cum_us[N + 1] - cum_us[N] is what op N adds to the production kernel: no stamps on the critical wave, and every workgroup under the load of all the
 * others (the profiling build's workgroup 0 runs in their wake).  cum_us[0] is the launch floor.  n must be nutls_fused_num_ops + 1.  Timing
 * only: the truncated launches leave the streams between two frames, every stream is reset afterwards (input: the library's mag_in buffer). */
int nutls_profile_production(nutls_handle* h, double* cum_us, int n, int reps, int steps);
/* Host-only (works without a GPU): the fused kernel's weight blob for a container -- conv kernels int8 in MFMA fragment
 * order, everything else fp32, in the order of the variant's static schedule.  n_floats must equal
 * nutls_fused_blob_floats(variant). */
int nutls_fused_blob_floats(int variant);
int nutls_fused_pack_blob(const void* weights, size_t n_bytes, int variant, float* out, size_t n_floats);
/* The same for the plan with `streams` streams per workgroup (1: the two entries above; 2: the packed plan of the LSTM variant, whose
 * tilings -- hence fragment order -- differ); nutls_fused_plan_blob_floats is 0 where no such plan exists. */
int nutls_fused_plan_blob_floats(int variant, int streams);
/* Op instances of that plan (a packed plan has one instance of a layer per group of streams that runs it side by side: "name#s2" = the
 * instance that starts at stream slot 2) -- the count nutls_profile_fused expects for a handle on that plan. */
int nutls_fused_plan_num_ops(int variant, int streams);
int nutls_fused_plan_op_info(int variant, int streams, int index, const char** name, double* flops);
int nutls_fused_pack_blob_plan(const void* weights, size_t n_bytes, int variant, int streams, float* out, size_t n_floats);

const char* nutls_last_error(void);
const char* nutls_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NUTLS_H_ */
