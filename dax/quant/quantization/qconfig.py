from inferix_amd.quant import (get_dynamic_fp8_per_tensor_act_per_channel_weight_qconfig,  # noqa: F401
                               get_dynamic_fp8_per_tensor_act_per_tensor_weight_qconfig,
                               get_dynamic_fp8_per_token_act_per_channel_weight_qconfig,
                               get_dynamic_int8_per_tensor_act_per_channel_weight_qconfig,
                               get_dynamic_int8_per_tensor_act_per_tensor_weight_qconfig,
                               get_dynamic_int8_per_token_act_per_channel_weight_qconfig)
