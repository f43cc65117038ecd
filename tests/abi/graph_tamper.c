/* Runtime-level drop-in, negative test: the reference's GENERATED network.c is compiled unchanged --
 * textually included here so that this translation unit can reach its static layer objects -- and ONE field of ONE layer
 * is changed in memory before ai_network_init.  The library must refuse to initialise (it implements exactly the yoloface
 * graph) instead of silently running its own graph.  Built only where /root/reference exists (oracle/Makefile.ref).
 *
 *   abi_graph_tamper <what>     what: none | stride | groups | pad | nl | pool | shape | order | weights | scale | zp | wscale | prescale
 * The quantisation records of network.c are const objects; the quantisation edits therefore hang an edited COPY of a record
 * on the tensor (its `klass` pointer), which is what a regenerated network.c with another calibration would look like.
 * prints "init ok" or "init failed type=.. code=.. : <text>".  Exit 0 = initialised, 4 = refused. */
#include <stdio.h>
#include <string.h>
#include "network.c"                   /* -I<reference>/stm32/X-CUBE-AI/App: the unmodified generated file */
#include "network_data.h"

extern const char* yf_network_last_error_text(ai_handle network);
AI_ALIGNED(32) static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE];

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "none";
  if (!strcmp(what, "stride")) conv2d_27_layer.filter_stride.data[0] = 1;                 /* stride 2x2 -> 1x2 */
  else if (!strcmp(what, "groups")) conv2d_15_layer.groups = 1;                            /* depthwise -> dense */
  else if (!strcmp(what, "pad")) ((ai_shape_dimension*)conv2d_1_layer.filter_pad.data)[2] = 1;   /* pad right as well */
  else if (!strcmp(what, "nl")) conv2d_5_layer.nl_func = nl_func_array_integer;            /* a LeakyReLU where there is none */
  else if (!strcmp(what, "pool")) pool_8_layer.pool_size.data[1] = 4;                      /* 8x8 window -> 8x4 */
  else if (!strcmp(what, "shape")) ((ai_shape_dimension*)conv2d_53_output.shape.data)[1] = 12;   /* 18 head channels -> 12 */
  else if (!strcmp(what, "order")) conv2d_3_layer.next = AI_NODE_OBJ(&conv2d_6_layer);     /* skip conv2d_5 */
  else if (!strcmp(what, "weights")) conv2d_47_weights_array.size = 1000;
  else if (!strcmp(what, "scale") || !strcmp(what, "zp") || !strcmp(what, "prescale")) {
    /* conv2d_23's output (scale), conv2d_30's input = conv2d_29's output (zero point), conv2d_47's pre-activation tensor */
    static ai_float sc[1]; static ai_i8 zp[1]; static ai_intq_info info[1] = {{ sc, (ai_handle)zp }}; static ai_intq_info_list lst;
    ai_tensor* t = !strcmp(what, "scale") ? &conv2d_23_output : !strcmp(what, "zp") ? &conv2d_29_output : &conv2d_47_scratch1;
    const ai_intq_info_list* q = (const ai_intq_info_list*)t->klass;
    sc[0] = q->info[0].scale[0]; zp[0] = ((const ai_i8*)q->info[0].zeropoint)[0];
    if (!strcmp(what, "zp")) zp[0] = (ai_i8)(zp[0] + 1); else sc[0] *= 1.0000002f;        /* one float32 ulp */
    lst.flags = q->flags; lst.size = 1; lst.info = info;
    t->klass = (ai_klass_obj)&lst;
  }
  else if (!strcmp(what, "wscale")) {           /* channel 17 of conv2d_6's filter scales */
    static ai_float sc[18]; static ai_i8 zp[18]; static ai_intq_info info[1] = {{ sc, (ai_handle)zp }}; static ai_intq_info_list lst;
    const ai_intq_info_list* q = (const ai_intq_info_list*)conv2d_6_weights.klass;
    for (int k = 0; k < 18; ++k) { sc[k] = q->info[0].scale[k]; zp[k] = 0; }
    sc[17] *= 1.0000002f;
    lst.flags = q->flags; lst.size = 18; lst.info = info;
    conv2d_6_weights.klass = (ai_klass_obj)&lst;
  }
  else if (strcmp(what, "none")) { fprintf(stderr, "unknown edit %s\n", what); return 2; }

  ai_handle network = AI_HANDLE_NULL;
  ai_error err = ai_network_create(&network, AI_NETWORK_DATA_CONFIG);
  if (err.type != AI_ERROR_NONE) { printf("create failed type=%d code=%d\n", err.type, err.code); return 3; }
  const ai_network_params params = AI_NETWORK_PARAMS_INIT(
      AI_NETWORK_DATA_WEIGHTS(ai_network_data_weights_get()),
      AI_NETWORK_DATA_ACTIVATIONS(activations));
  if (!ai_network_init(network, &params)) {
    err = ai_network_get_error(network);
    printf("init failed type=0x%x code=0x%x : %s\n", err.type, err.code, yf_network_last_error_text(network));
    return 4;
  }
  printf("init ok\n");
  ai_network_destroy(network);
  return 0;
}
