/* C host layer of libyf_network.so: the reference's ai_network_* boundary re-implemented over the HIP engine.
 *
 * Mirrors the generated reference file stm32/X-CUBE-AI/App/network.c:3270-3413 (public APIs) and
 * network_data.c:393-432 (weights/params getters): same entry points, same argument meaning, same
 * first-error-latch convention (network.h:120-132), one static context (network.c:2929-2939 `g_network`).
 * What the reference does on the MCU with pointer binding (network_configure_weights/activations,
 * network.c:2943-3267) becomes: read the caller's weight blob, build device tables, upload to HBM.
 *
 * There is NO CPU compute path in this file or behind it: without a gfx950 GPU ai_network_init fails with
 * AI_ERROR_INIT_FAILED and ai_network_run returns 0.
 */
#define _GNU_SOURCE            /* RTLD_DEFAULT, RTLD_NOLOAD */
#include "../../include/yf_network.h"
#include "yf_engine.h"
#include "yf_host_prep.h"
#include "yf_impl.h"
#include "yf_fp16.h"
#include "gen/yf_model_gen.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern const uint8_t yf_weights_blob[AI_NETWORK_DATA_WEIGHTS_SIZE];   /* gen/yf_weights_blob_gen.c */

enum { ST_NONE = 0, ST_CREATED = 1, ST_READY = 2 };

typedef struct {
  int state;
  ai_error first_error;
  yf_engine* engine;
  yf_fp16* fp16;
  int device;
  int cfg_frames, cfg_waves;
  int rounding;                           /* YF_ROUND_*: which published rounding the requantisation constants are built for */
  size_t bound_weights_bytes;
  ai_platform_version tools_api;          /* what the network was created with (network.c:3376: AI_TOOLS_API_VERSION_*); reports return it */
  const void* bound_weights;              /* what ai_network_init was handed: the reports describe these buffers */
  ai_handle bound_activations;
  char err_text[512];
} yf_context;

static yf_context g_network;             /* the reference has exactly one instance too (network.c:35-36) */

static yf_context* acquire(ai_handle h) { return (h == (ai_handle)&g_network && g_network.state != ST_NONE) ? &g_network : NULL; }

static void latch(yf_context* c, unsigned type, unsigned code, const char* text) {
  if (c->first_error.type == AI_ERROR_NONE) { c->first_error.type = type; c->first_error.code = code; }
  if (text) snprintf(c->err_text, sizeof c->err_text, "%s", text);
}

void yf_impl_fail_init(ai_handle network, unsigned code, const char* text) {
  yf_context* c = acquire(network);
  if (c) latch(c, AI_ERROR_INIT_FAILED, code, text);
}

static ai_platform_version version3(unsigned a, unsigned b, unsigned c);
static ai_error mk_error(unsigned type, unsigned code) { ai_error e; e.type = type; e.code = code; return e; }

/* $YF_REQUANT_ROUNDING: lets an unmodified aiInit() (yoloface.c:188-211: create + init back to back) be steered from outside */
static int rounding_from_env(void) {
  const char* v = getenv("YF_REQUANT_ROUNDING");
  if (!v || !*v) return YF_ROUND_TFLITE_REF;
  char word[32];
  snprintf(word, sizeof word, "%s", v);
  int flags = 0;
  char* plus = strchr(word, '+');          /* "ties_up+generic": that rounding on the four-instruction kernels (A/B of the two kernel sets) */
  if (plus) { if (strcmp(plus, "+generic")) return -1; *plus = 0; flags = YF_ROUND_GENERIC_KERNELS; }
  if (!strcmp(word, "ref")) return YF_ROUND_TFLITE_REF | flags;
  if (!strcmp(word, "ties_up")) return YF_ROUND_TIES_UP | flags;
  if (!strcmp(word, "ties_up_all")) return YF_ROUND_TIES_UP_ALL | flags;
  if (!strcmp(word, "single")) return YF_ROUND_SINGLE | flags;
  return -1;                               /* unknown word: ai_network_init fails loudly instead of guessing */
}
static int rounding_is_valid(int r) { return r >= 0 && (r & ~YF_ROUND_GENERIC_KERNELS) < YF_ROUND_COUNT; }

/* ------------------------------------------------------------------------------------------------ create / destroy */
ai_error yf_impl_create(ai_handle* network, const ai_buffer* network_config) {
  if (!network) return mk_error(AI_ERROR_CREATE_FAILED, AI_ERROR_CODE_INVALID_PTR);
  if (network_config != NULL) { *network = AI_HANDLE_NULL; return mk_error(AI_ERROR_CREATE_FAILED, AI_ERROR_CODE_NETWORK); }
  if (g_network.state == ST_READY && g_network.engine) { yf_engine_destroy(g_network.engine); }
  if (g_network.state != ST_NONE && g_network.fp16) { yf_fp16_destroy(g_network.fp16); }
  const int dev = g_network.state != ST_NONE ? g_network.device : 0;
  const int cf = g_network.cfg_frames, cw = g_network.cfg_waves;
  memset(&g_network, 0, sizeof g_network);
  g_network.state = ST_CREATED;
  g_network.device = dev; g_network.cfg_frames = cf; g_network.cfg_waves = cw;
  g_network.tools_api = version3(YF_REPORT_TOOLS_API_VERSION);
  g_network.rounding = rounding_from_env();
  *network = (ai_handle)&g_network;
  return mk_error(AI_ERROR_NONE, AI_ERROR_CODE_NONE);
}

ai_handle yf_impl_destroy(ai_handle network) {
  yf_context* c = acquire(network);
  if (!c) return network;                       /* not destroyed: same handle comes back (network.h:155-157) */
  if (c->engine) yf_engine_destroy(c->engine);
  if (c->fp16) yf_fp16_destroy(c->fp16);
  memset(c, 0, sizeof *c);
  return AI_HANDLE_NULL;
}

ai_error yf_impl_get_error(ai_handle network) {
  yf_context* c = acquire(network);
  if (!c) return mk_error(AI_ERROR_INVALID_HANDLE, AI_ERROR_CODE_NETWORK);
  const ai_error e = c->first_error;
  c->first_error = mk_error(AI_ERROR_NONE, AI_ERROR_CODE_NONE);
  return e;
}

/* ------------------------------------------------------------------------------------------------ init */
static size_t buffer_elems(const ai_buffer* b) { return (size_t)b->height * b->width * b->channels; }

/* The weights arrive either as the legacy {params, activations} pair whose params.data points at the
 * {AI_MAGIC_MARKER, blob, AI_MAGIC_MARKER} pointer map (network_data.c:395-401, used by yoloface.c:199-202), or as
 * the signed ai_buffer_array map produced by ai_network_data_params_get (network_data.c:412-432). */
const uint8_t* yf_impl_resolve_weights(const ai_network_params* p, size_t* bytes, const ai_buffer** act) {
  if (p->map_signature == (ai_signature)AI_MAGIC_SIGNATURE) {
    if (p->map_weights.size < 1 || !p->map_weights.buffer) return NULL;
    const ai_buffer* wb = &p->map_weights.buffer[0];
    *bytes = buffer_elems(wb);
    *act = (p->map_activations.size >= 1) ? &p->map_activations.buffer[0] : NULL;
    return (const uint8_t*)wb->data;
  }
  const ai_buffer* wb = &p->params;
  *bytes = buffer_elems(wb);
  *act = &p->activations;
  if (!wb->data) return NULL;
  const uintptr_t* map = (const uintptr_t*)wb->data;
  if (map[0] == (uintptr_t)AI_MAGIC_MARKER) {
    if (map[2] != (uintptr_t)AI_MAGIC_MARKER) return NULL;
    return (const uint8_t*)map[1];
  }
  return (const uint8_t*)wb->data;                /* bare blob pointer */
}

ai_bool yf_impl_init(ai_handle network, const ai_network_params* params) {
  yf_context* c = acquire(network);
  if (!c) return false;
  if (!params) { latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_PARAMS, "params is NULL"); return false; }
  size_t wbytes = 0;
  const ai_buffer* act = NULL;
  const uint8_t* blob = yf_impl_resolve_weights(params, &wbytes, &act);
  if (!blob) { latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_WEIGHTS, "weights buffer/map is invalid"); return false; }
  if (wbytes < AI_NETWORK_DATA_WEIGHTS_SIZE) { latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_INVALID_SIZE, "weights buffer smaller than 11304 bytes"); return false; }
  /* the activations arena is caller-owned scratch on the MCU; here activations live in LDS.  It is accepted and
   * left untouched, but a too-small arena is still reported the way the reference runtime would. */
  if (act && act->data && buffer_elems(act) < AI_NETWORK_DATA_ACTIVATIONS_SIZE) {
    latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_ACTIVATIONS, "activations buffer smaller than 29784 bytes"); return false;
  }
  if (c->engine) { yf_engine_destroy(c->engine); c->engine = NULL; c->state = ST_CREATED; }
  if (!rounding_is_valid(c->rounding)) {
    latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_PARAMS, "YF_REQUANT_ROUNDING is none of ref, ties_up, ties_up_all, single (optionally followed by +generic)"); return false;
  }

  uint8_t* tables = NULL;
  yf_table_index ix;
  const int prc = yf_prepare_tables_rounding(blob, wbytes, c->rounding, &tables, &ix);
  if (prc != YF_PREP_OK) {
    char t[96]; snprintf(t, sizeof t, "table preparation failed (code %d)", prc);
    latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_WEIGHTS, t); return false;
  }
  char etext[400] = "";
  const int erc = yf_engine_create(c->device, tables, &ix, yf_rounding_signless_dense(c->rounding), &c->engine, etext, sizeof etext);
  free(tables);
  if (erc != YF_ENG_OK) { c->engine = NULL; latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK, etext); return false; }
  if (c->cfg_frames || c->cfg_waves) {
    if (yf_engine_configure(c->engine, c->cfg_frames, c->cfg_waves) != YF_ENG_OK) {
      latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_PARAMS, yf_engine_error(c->engine));
      yf_engine_destroy(c->engine); c->engine = NULL; return false;
    }
  }
  c->state = ST_READY;
  c->bound_weights = blob;
  c->bound_weights_bytes = wbytes;
  c->bound_activations = act ? act->data : NULL;
  return true;
}

/* ------------------------------------------------------------------------------------------------ run */
static int check_io(yf_context* c, const ai_buffer* b, int is_input) {
  const unsigned etype = is_input ? AI_ERROR_INVALID_INPUT : AI_ERROR_INVALID_OUTPUT;
  if (!b || !b->data) { latch(c, etype, AI_ERROR_CODE_INVALID_PTR, is_input ? "input buffer/data is NULL" : "output buffer/data is NULL"); return 0; }
  /* compare type/sign/bits, ignore the flag bits 24..30 (CONST/STATIC/IS_IO) */
  if ((b->format & 0x00FFFFFF) != (AI_BUFFER_FORMAT_S8 & 0x00FFFFFF)) { latch(c, etype, AI_ERROR_CODE_INVALID_FORMAT, "buffer format is not S8"); return 0; }
  const int h = is_input ? AI_NETWORK_IN_1_HEIGHT : AI_NETWORK_OUT_1_HEIGHT, w = is_input ? AI_NETWORK_IN_1_WIDTH : AI_NETWORK_OUT_1_WIDTH;
  const unsigned ch = is_input ? AI_NETWORK_IN_1_CHANNEL : AI_NETWORK_OUT_1_CHANNEL;
  if (b->height != h || b->width != w || b->channels != ch) { latch(c, etype, AI_ERROR_CODE_INVALID_SIZE, "buffer shape mismatch"); return 0; }
  if (b->n_batches < 1) { latch(c, etype, AI_ERROR_CODE_INVALID_BATCH, "n_batches < 1"); return 0; }
  return 1;
}

static ai_i32 process(ai_handle network, const ai_buffer* input, ai_buffer* output) {
  yf_context* c = acquire(network);
  if (!c) return 0;
  if (c->state != ST_READY || !c->engine) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_MISSED_INIT, "network not initialised"); return 0; }
  if (!check_io(c, input, 1)) return 0;
  if (output) {
    if (!check_io(c, output, 0)) return 0;
    if (output->n_batches < input->n_batches) { latch(c, AI_ERROR_INVALID_OUTPUT, AI_ERROR_CODE_INVALID_BATCH, "output holds fewer batches than input"); return 0; }
  }
  const long n = input->n_batches;
  const int rc = yf_engine_run_host(c->engine, input->data, output ? output->data : NULL, n);
  if (rc != YF_ENG_OK) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, yf_engine_error(c->engine)); return 0; }
  return (ai_i32)n;
}

ai_i32 yf_impl_run(ai_handle network, const ai_buffer* input, ai_buffer* output) {
  yf_context* c = acquire(network);
  if (c && !output) { latch(c, AI_ERROR_INVALID_OUTPUT, AI_ERROR_CODE_INVALID_PTR, "output is NULL"); return 0; }
  return process(network, input, output);
}

ai_i32 yf_impl_forward(ai_handle network, const ai_buffer* input) { return process(network, input, NULL); }

/* Per-node observer (platform_abi.c): validates like ai_network_run, then runs the debug build on frames [first, first + count) of
 * `input`: heads into `heads` (count x 882 bytes), every node's tensor into `dump` (count x yf_impl_dump_bytes()).  0 on error (latched). */
ai_i32 yf_impl_run_dump(ai_handle network, const ai_buffer* input, const ai_buffer* output, long first, long count, int8_t* heads, int8_t* dump) {
  yf_context* c = acquire(network);
  if (!c) return 0;
  if (c->state != ST_READY || !c->engine) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_MISSED_INIT, "network not initialised"); return 0; }
  if (!check_io(c, input, 1)) return 0;
  if (output) {
    if (!check_io(c, output, 0)) return 0;
    if (output->n_batches < input->n_batches) { latch(c, AI_ERROR_INVALID_OUTPUT, AI_ERROR_CODE_INVALID_BATCH, "output holds fewer batches than input"); return 0; }
  }
  if (first < 0 || count < 1 || first + count > input->n_batches || !heads || !dump) { latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_OUT_OF_RANGE, "observed frame range"); return 0; }
  const int rc = yf_engine_run_host_dump(c->engine, (const int8_t*)input->data + first * AI_NETWORK_IN_1_SIZE, heads, dump, count);
  if (rc != YF_ENG_OK) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, yf_engine_error(c->engine)); return 0; }
  return (ai_i32)count;
}
long yf_impl_dump_bytes(void) { return yf_engine_dump_bytes(); }
long yf_impl_dump_offset(int tflite_op) { return yf_engine_dump_offset(tflite_op); }
void yf_impl_fail_run(ai_handle network, unsigned code, const char* text) {
  yf_context* c = acquire(network);
  if (c) latch(c, AI_ERROR_INVALID_STATE, code, text);
}

/* ------------------------------------------------------------------------------------------------ report */
static ai_buffer g_io_in = { AI_BUFFER_FORMAT_S8, 1, AI_NETWORK_IN_1_HEIGHT, AI_NETWORK_IN_1_WIDTH, AI_NETWORK_IN_1_CHANNEL, NULL, NULL };
static ai_buffer g_io_out = { AI_BUFFER_FORMAT_S8, 1, AI_NETWORK_OUT_1_HEIGHT, AI_NETWORK_OUT_1_WIDTH, AI_NETWORK_OUT_1_CHANNEL, NULL, NULL };
static ai_buffer g_rep_weights[1], g_rep_activations[1];     /* what a report's map_weights / map_activations point at */

static ai_platform_version version3(unsigned a, unsigned b, unsigned c) { ai_platform_version v; v.major = (ai_u8)a; v.minor = (ai_u8)b; v.micro = (ai_u8)c; v.reserved = 0; return v; }
const char* yf_impl_runtime_revision(void) { return "yf-mi355x (gfx950 fused int8 engine)"; }
ai_platform_version yf_impl_runtime_version(void) { return version3(0, 1, 0); }
ai_platform_version yf_impl_api_version(void) { return version3(YF_REPORT_PLATFORM_API_VERSION); }          /* AI_PLATFORM_API_VERSION, network_config.h:33-38 */
ai_platform_version yf_impl_interface_api_version(void) { return version3(1, 3, 0); }                       /* AI_PLATFORM_INTERFACE_API 1.3.0 of the 7.0.0 runtime */

void yf_impl_set_tools_api_version(ai_handle network, unsigned major, unsigned minor, unsigned micro) {
  yf_context* c = acquire(network);
  if (c) c->tools_api = version3(major, minor, micro);
}

/* The RUNTIME's half of a report (the closed library's ai_platform_api_get_network_report, ai_platform_interface.h:818-825, which the reference's
 * network.c:3307,3352 calls on a report it has pre-filled): I/O descriptors, node count, the tools API version the network was created with, and the
 * buffer description in the arm of the union the caller asked for -- ai_network_get_report pre-sets map_signature = AI_MAGIC_SIGNATURE
 * (network.c:3346) and gets map_weights / map_activations, the deprecated ai_network_get_info leaves it 0 (network.c:3301-3302) and gets the legacy
 * params / activations pair.  data = what ai_network_init was given (NULL before it). */
ai_bool yf_impl_fill_report(ai_handle network, ai_network_report* r) {
  yf_context* c = acquire(network);
  if (!c || !r) return false;
  const ai_buffer w = { (ai_buffer_format)(AI_BUFFER_FORMAT_U8 | AI_BUFFER_FMT_FLAG_CONST), 1, 1, 1, AI_NETWORK_DATA_WEIGHTS_SIZE, (ai_handle)c->bound_weights, NULL };
  const ai_buffer a = { AI_BUFFER_FORMAT_U8, 1, 1, 1, AI_NETWORK_DATA_ACTIVATIONS_SIZE, c->bound_activations, NULL };
  r->tool_api_version = c->tools_api;
  r->n_inputs = 1; r->n_outputs = 1;
  r->inputs = &g_io_in; r->outputs = &g_io_out;
  if (r->map_signature == (ai_signature)AI_MAGIC_SIGNATURE) {
    g_rep_weights[0] = w; g_rep_activations[0] = a;
    r->map_weights.flags = 0; r->map_weights.size = 1; r->map_weights.buffer = g_rep_weights;
    r->map_activations.flags = 0; r->map_activations.size = 1; r->map_activations.buffer = g_rep_activations;
  } else {
    r->params = w; r->activations = a;
  }
  r->n_nodes = AI_NETWORK_N_NODES;
  r->signature = 0;
  return true;
}

/* network.c's half (network.c:3271-3361): the identity of the generated model, as the #defines of the reference's generated files give it
 * (gen/yf_model_gen.h, written by tools/gen_model.py).  compile_datetime is this library's. */
static ai_bool report(ai_handle network, ai_network_report* out, int buffer_maps) {
  yf_context* c = acquire(network);
  if (!c || !out) return false;
  ai_network_report r;
  memset(&r, 0, sizeof r);
  r.model_name = YF_REPORT_MODEL_NAME;
  r.model_signature = YF_REPORT_MODEL_SIGNATURE;
  r.model_datetime = YF_REPORT_MODEL_DATETIME;
  r.compile_datetime = __DATE__ " " __TIME__;
  r.runtime_revision = yf_impl_runtime_revision();
  r.runtime_version = yf_impl_runtime_version();
  r.tool_revision = YF_REPORT_TOOLS_REVISION_ID;
  r.tool_version = version3(YF_REPORT_TOOLS_VERSION);
  r.api_version = yf_impl_api_version();
  r.interface_api_version = yf_impl_interface_api_version();
  r.n_macc = YF_REPORT_N_MACC;                           /* network.c:3298, network_generate_report.txt:20 */
  if (buffer_maps) r.map_signature = (ai_signature)AI_MAGIC_SIGNATURE;
  if (!yf_impl_fill_report(network, &r)) return false;
  *out = r;
  return true;
}
ai_bool yf_impl_get_report(ai_handle network, ai_network_report* r) { return report(network, r, 1); }
ai_bool yf_impl_get_info(ai_handle network, ai_network_report* r) { return report(network, r, 0); }

/* ------------------------------------------------------------------------------------------------ public boundary
 * Thin wrappers: the implementations above carry private names so that the runtime-level entry points
 * (platform_abi.c) can reach them even when an application links the reference's own network.c, whose
 * ai_network_* definitions then take precedence over the ones exported here. */
YF_API ai_error  ai_network_create(ai_handle* network, const ai_buffer* network_config) { return yf_impl_create(network, network_config); }
YF_API ai_handle ai_network_destroy(ai_handle network) { return yf_impl_destroy(network); }
YF_API ai_error  ai_network_get_error(ai_handle network) { return yf_impl_get_error(network); }
YF_API ai_bool   ai_network_init(ai_handle network, const ai_network_params* params) { return yf_impl_init(network, params); }
YF_API ai_i32    ai_network_run(ai_handle network, const ai_buffer* input, ai_buffer* output) { return yf_impl_run(network, input, output); }
YF_API ai_i32    ai_network_forward(ai_handle network, const ai_buffer* input) { return yf_impl_forward(network, input); }
YF_API ai_bool   ai_network_get_report(ai_handle network, ai_network_report* report) { return yf_impl_get_report(network, report); }
YF_API ai_bool   ai_network_get_info(ai_handle network, ai_network_report* report) { return yf_impl_get_info(network, report); }

/* ------------------------------------------------------------------------------------------------ network_data */
YF_API ai_handle ai_network_data_weights_get(void) {
  static const uint8_t* map[3];
  map[0] = (const uint8_t*)(uintptr_t)AI_MAGIC_MARKER;
  map[1] = yf_weights_blob;
  map[2] = (const uint8_t*)(uintptr_t)AI_MAGIC_MARKER;
  return AI_HANDLE_PTR(map);
}

YF_API ai_bool ai_platform_bind_network_params(ai_handle network, ai_network_params* params,
                                               const ai_buffer_array* map_weights, const ai_buffer_array* map_activations) {
  if (!network || !params || !map_weights || !map_activations) return false;
  memset(params, 0, sizeof *params);
  params->map_signature = (ai_signature)AI_MAGIC_SIGNATURE;
  params->map_weights = *map_weights;
  params->map_activations = *map_activations;
  return true;
}

YF_API ai_bool ai_network_data_params_get(ai_handle network, ai_network_params* params) {
  if (!(network && params)) return false;
  static ai_buffer act[1] = {{ AI_BUFFER_FORMAT_U8, 1, 1, 1, AI_NETWORK_DATA_ACTIVATIONS_SIZE, NULL, NULL }};
  static ai_buffer wts[1] = {{ AI_BUFFER_FORMAT_U8, 1, 1, 1, AI_NETWORK_DATA_WEIGHTS_SIZE, NULL, NULL }};
  wts[0].data = (ai_handle)yf_weights_blob;
  const ai_buffer_array ma = { 0, 1, act };
  const ai_buffer_array mw = { 0, 1, wts };
  return ai_platform_bind_network_params(network, params, &mw, &ma);
}

/* ------------------------------------------------------------------------------------------------ extensions */
YF_API int yf_network_set_device(ai_handle network, int device) {
  yf_context* c = acquire(network);
  if (!c || device < 0) return -1;
  if (c->state == ST_READY) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_IN_USE, "set the device before ai_network_init"); return -1; }
  c->device = device;
  return 0;
}

YF_API int yf_network_set_requant_rounding(ai_handle network, int rounding) {
  yf_context* c = acquire(network);
  if (!c) return -1;
  if (!rounding_is_valid(rounding)) { latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_OUT_OF_RANGE, "no such requantisation rounding"); return -1; }
  if (c->state == ST_READY && c->engine && rounding != c->rounding) {
    /* same weights (still the caller's, as on the MCU: network.c:3108-3267 keeps pointers into the blob), other constants, same layout */
    uint8_t* tables = NULL;
    yf_table_index ix;
    const int prc = yf_prepare_tables_rounding((const uint8_t*)c->bound_weights, c->bound_weights_bytes, rounding, &tables, &ix);
    if (prc != YF_PREP_OK) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK_WEIGHTS, "table preparation failed"); return -1; }
    const int erc = yf_engine_set_tables(c->engine, tables, &ix, yf_rounding_signless_dense(rounding));
    free(tables);
    if (erc != YF_ENG_OK) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, yf_engine_error(c->engine)); return -1; }
  }
  c->rounding = rounding;
  return 0;
}
YF_API int yf_network_get_requant_rounding(ai_handle network) {
  yf_context* c = acquire(network);
  return c ? c->rounding : -1;
}

YF_API int yf_network_configure(ai_handle network, int frames_per_wg, int waves_per_wg) {
  yf_context* c = acquire(network);
  if (!c) return -1;
  if (frames_per_wg < 0) {                       /* automatic: the shipped throughput shape, small batches one frame per workgroup */
    if (c->state == ST_READY) (void)yf_engine_configure(c->engine, -1, -1);
    c->cfg_frames = c->cfg_waves = 0;
    return 0;
  }
  /* validate first: a rejected shape must not poison the stored configuration (every later ai_network_init would fail) */
  const int f = frames_per_wg > 0 ? frames_per_wg : (c->cfg_frames > 0 ? c->cfg_frames : 2);
  const int w = waves_per_wg > 0 ? waves_per_wg : (c->cfg_waves > 0 ? c->cfg_waves : 8);
  if (!yf_engine_variant_exists(f, w)) { latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_NETWORK_PARAMS, "no such kernel variant"); return -1; }
  if (c->state == ST_READY && yf_engine_configure(c->engine, f, w) != YF_ENG_OK) {
    latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_NETWORK_PARAMS, yf_engine_error(c->engine)); return -1;
  }
  c->cfg_frames = f; c->cfg_waves = w;
  return 0;
}

static yf_context* ready(ai_handle network) {
  yf_context* c = acquire(network);
  if (!c) return NULL;
  if (c->state != ST_READY || !c->engine) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_MISSED_INIT, "network not initialised"); return NULL; }
  return c;
}

static long finish(yf_context* c, int rc, long n) {
  if (rc == YF_ENG_OK) return n;
  latch(c, rc == YF_ENG_ERR_ARG ? AI_ERROR_INVALID_PARAM : AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, yf_engine_error(c->engine));
  return 0;
}

YF_API long yf_network_run_device(ai_handle network, const void* d_in, void* d_out, long n, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_run_device(c->engine, d_in, d_out, NULL, n, stream), n);
}

YF_API long yf_network_run_device_dump(ai_handle network, const void* d_in, void* d_out, void* d_dump, long n, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  if (!d_dump) { latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_INVALID_PTR, "dump buffer is NULL"); return 0; }
  return finish(c, yf_engine_run_device(c->engine, d_in, d_out, d_dump, n, stream), n);
}

YF_API long yf_network_dump_bytes(void) { return yf_engine_dump_bytes(); }

YF_API long yf_network_run_device_hw(ai_handle network, int height, int width, const void* d_in, void* d_out, long n, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  if (height == 56 && width == 56) return finish(c, yf_engine_run_device(c->engine, d_in, d_out, NULL, n, stream), n);
  if (height == 160 && width == 160) return finish(c, yf_engine_run_device_160(c->engine, d_in, d_out, n, stream), n);
  latch(c, AI_ERROR_INVALID_INPUT, AI_ERROR_CODE_INVALID_SIZE, "supported input sizes: 56x56 (one fused kernel) and 160x160 (banded kernels)");
  return 0;
}

YF_API int yf_network_table_plan(int32_t* out, int cap) { return yf_engine_table_plan(out, cap); }

/* ------------------------------------------------------------------------------------------------ multi-GPU */
YF_API void yf_network_shard_range(long n, int rank, int world, long* begin, long* end) {
  if (world < 1) world = 1;
  if (rank < 0) rank = 0;
  const long base = n / world, extra = n % world;
  const long b = rank * base + (rank < extra ? rank : extra);
  if (begin) *begin = b;
  if (end) *end = b + base + (rank < extra ? 1 : 0);
}

/* ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream) out of librccl.so, resolved on first use */
typedef int (*yf_nccl_all_gather_fn)(const void*, void*, size_t, int, void*, void*);
static yf_nccl_all_gather_fn g_rccl_all_gather;
static pthread_once_t g_rccl_once = PTHREAD_ONCE_INIT;
static void resolve_rccl(void) {
  void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
  if (!sym) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) sym = dlsym(h, "ncclAllGather");
  }
  g_rccl_all_gather = (yf_nccl_all_gather_fn)sym;
}
YF_API long yf_network_all_gather_device(ai_handle network, void* nccl_comm, const void* d_send, void* d_recv,
                                         size_t bytes_per_rank, void* stream) {
  yf_context* c = acquire(network);
  if (!c) return 0;
  if (!nccl_comm || !d_send || !d_recv || bytes_per_rank == 0) { latch(c, AI_ERROR_INVALID_PARAM, AI_ERROR_CODE_INVALID_PTR, "all-gather: NULL communicator/buffer or zero size"); return 0; }
  /* Resolve ncclAllGather in the RCCL instance that created the caller's communicator: first whatever the process already
   * has (global scope, then an already loaded librccl.so.1 -- glibc matches loaded objects by SONAME, and both the ROCm and
   * the PyTorch copy carry that one), only then load a copy.  Resolved once (pthread_once). */
  pthread_once(&g_rccl_once, resolve_rccl);
  const yf_nccl_all_gather_fn fn = g_rccl_all_gather;
  if (!fn) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, "all-gather: librccl.so.1 / ncclAllGather not found"); return 0; }
  const int rc = fn(d_send, d_recv, bytes_per_rank, 0 /* ncclInt8 */, nccl_comm, stream);
  if (rc != 0) {
    char t[96]; snprintf(t, sizeof t, "ncclAllGather failed with ncclResult_t %d", rc);
    latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, t); return 0;
  }
  return (long)bytes_per_rank;
}

/* stm32/User/main.c:46,53 + yoloface.c:148: the text the firmware prints per frame, byte for byte. */
YF_API long yf_network_format_uart(unsigned frame_no, const yf_det* dets, int count, int cap, char* buf, size_t buflen) {
  static const char dashes[] = "----------------------------------------";
  long need = 0;
  size_t room = buf ? buflen : 0;
#define YF_EMIT(...)                                                                                              \
  do {                                                                                                            \
    const size_t at_ = (size_t)need < room ? (size_t)need : room;                                                 \
    const int w_ = snprintf(room ? buf + at_ : NULL, room - at_, __VA_ARGS__);                                    \
    if (w_ < 0) { return -1; }                                                                                    \
    need += w_;                                                                                                   \
  } while (0)
  YF_EMIT("=== Frame %d ===\r\n%s\r\n", (int)frame_no, dashes);
  const int lines = dets ? (count < cap ? count : cap) : 0;
  for (int k = 0; k < lines; ++k)        /* face_num is a uint8_t counted before the print (yoloface.c:125) */
    YF_EMIT("[Face %d] BBox: [%d, %d, %d, %d], Conf: %.2f\r\n", (int)(uint8_t)(k + 1), dets[k].x1, dets[k].y1, dets[k].x2, dets[k].y2,
            (double)dets[k].conf);
  YF_EMIT("%s\r\n[INFO] Total faces detected: %d\r\n", dashes, (int)(uint8_t)count);
#undef YF_EMIT
  return need;
}

YF_API long yf_network_decode_device(ai_handle network, const void* d_heads, long n, int mode, float w_scale, float h_scale,
                                     void* d_dets, void* d_counts, int cap, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_decode_device(c->engine, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream), n);
}

YF_API long yf_network_pack_detections_device(ai_handle network, const void* d_dets, const void* d_counts, const void* d_heads, void* d_wire, long n, int cap, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_pack_detections_device(c->engine, d_dets, d_counts, d_heads, d_wire, n, cap, stream), n);
}

YF_API long yf_network_unpack_detections_device(ai_handle network, const void* d_wire, const void* d_counts, void* d_heads, long n, int cap, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_unpack_detections_device(c->engine, d_wire, d_counts, d_heads, n, cap, stream), n);
}

YF_API long yf_network_run_decode_device(ai_handle network, const void* d_in, void* d_heads, long n, int mode, float w_scale, float h_scale,
                                         void* d_dets, void* d_counts, int cap, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_run_decode_device(c->engine, d_in, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream), n);
}

YF_API long yf_network_prepare_rgb565_device(ai_handle network, const void* d_rgb565, void* d_out, long n, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_prepare_rgb565_device(c->engine, d_rgb565, d_out, n, stream), n);
}

YF_API long yf_network_run_camera_device(ai_handle network, const void* d_rgb565, void* d_heads, long n, int mode, float w_scale, float h_scale,
                                         void* d_dets, void* d_counts, int cap, void* stream) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_run_camera_device(c->engine, d_rgb565, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream), n);
}

YF_API long yf_network_time_device(ai_handle network, const void* d_in, void* d_out, long n, int iters, void* stream, float* ms_per_launch) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_time_device(c->engine, d_in, d_out, n, iters, stream, ms_per_launch), n);
}

YF_API long yf_network_time_stages(ai_handle network, const void* d_in, void* d_out, long n, int iters, int stop_stage, void* stream, float* ms_per_launch) {
  yf_context* c = ready(network);
  if (!c) return 0;
  return finish(c, yf_engine_time_stages(c->engine, d_in, d_out, n, iters, stop_stage, stream, ms_per_launch), n);
}

/* fp16 side configuration: independent of ai_network_init (different weights: the fp32 ONNX export) */
YF_API int yf_network_fp16_init(ai_handle network, const void* yfw, size_t bytes) {
  yf_context* c = acquire(network);
  if (!c) return -1;
  if (c->fp16) { yf_fp16_destroy(c->fp16); c->fp16 = NULL; }
  char etext[300] = "";
  if (yf_fp16_create(c->device, yfw, bytes, &c->fp16, etext, sizeof etext) != 0) {
    c->fp16 = NULL; latch(c, AI_ERROR_INIT_FAILED, AI_ERROR_CODE_NETWORK_WEIGHTS, etext); return -1;
  }
  return 0;
}

YF_API long yf_network_fp16_run_device(ai_handle network, const void* d_in_f16, void* d_out_f32, long n, void* stream) {
  yf_context* c = acquire(network);
  if (!c) return 0;
  if (!c->fp16) { latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_MISSED_INIT, "yf_network_fp16_init first"); return 0; }
  if (yf_fp16_run_device(c->fp16, d_in_f16, d_out_f32, n, stream) != 0) {
    latch(c, AI_ERROR_INVALID_STATE, AI_ERROR_CODE_NETWORK, yf_fp16_error(c->fp16)); return 0;
  }
  return n;
}

/* Scratch regions (tail-batching slots, the fp16 park slots, the 160x160 arena) belong to the launch STREAM and are bounded (at most eight per kind,
 * recycled once their last launch has completed).  A caller that destroys a stream hands its regions back first. */
YF_API int yf_network_release_stream(ai_handle network, void* stream) {
  yf_context* c = acquire(network);
  if (!c || !c->engine) return -1;
  int rc = yf_engine_release_stream(c->engine, stream);
  if (c->fp16 && yf_fp16_release_stream(c->fp16, stream) != 0) rc = -1;
  return rc == 0 ? 0 : -1;
}
YF_API size_t yf_network_scratch_bytes(ai_handle network) {
  yf_context* c = acquire(network);
  if (!c || !c->engine) return 0;
  return yf_engine_scratch_bytes(c->engine) + (c->fp16 ? yf_fp16_scratch_bytes(c->fp16) : 0);
}

YF_API int yf_network_scratch_stats(ai_handle network, yf_scratch_stats* out) {
  yf_context* c = acquire(network);
  if (!c || !out) return -1;
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
  if (c->engine) yf_engine_scratch_stats(c->engine, v);
  if (c->fp16) yf_fp16_scratch_stats(c->fp16, v);
  out->events_recorded = v[0]; out->events_skipped = v[1]; out->event_waits = v[2]; out->device_syncs = v[3]; out->acquire_waits = v[4]; out->regions = v[5];
  return 0;
}

YF_API const char* yf_network_last_error_text(ai_handle network) {
  yf_context* c = acquire(network);
  /* with the reference's generated network.c in front (runtime-level path) the caller's handle is ITS ai_network object,
   * not this library's context: there is one instance either way, so any non-NULL handle reads the singleton's text */
  if (!c && network != AI_HANDLE_NULL && g_network.state != ST_NONE) c = &g_network;
  return c ? c->err_text : "invalid handle";
}

YF_API const char* yf_network_build_id(void) { return yf_engine_build_id(); }
#ifndef YF_HOST_ID
#define YF_HOST_ID "unknown"
#endif
YF_API const char* yf_network_host_id(void) { return YF_HOST_ID; }

YF_API const char* yf_network_kernel_name(ai_handle network) {
  yf_context* c = acquire(network);
  return (c && c->engine) ? yf_engine_kernel_name(c->engine) : "";
}

YF_API const char* yf_network_kernel_name_for(ai_handle network, long n) {
  yf_context* c = acquire(network);
  return (c && c->engine) ? yf_engine_kernel_name_for(c->engine, n) : "";
}
