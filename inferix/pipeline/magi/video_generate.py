"""inferix/pipeline/magi/video_generate.py: the planning and loop half of SampleTransport (:166-668) — generate_sequences, init_t,
init_intervel, the per-step plan, key ranges, forward_velocity + integrate_velocity as ChunkSchedule.run"""
from inferix_amd.magi.kv_ranges import (chunk_token_nums, generate_default_kvrange, generate_kvrange_for_denoising_video,  # noqa: F401
                                        generate_kvrange_for_prefix_video, generate_noise2clean_kvrange)
from inferix_amd.magi.schedule import ChunkSchedule, ForwardPlan, generate_sequences, init_t  # noqa: F401
from inferix_amd.magi.schedule import init_interval as init_intervel  # noqa: F401  (the reference's spelling)


def find_dit_model(model):
    """:246-251"""
    if hasattr(model, "y_embedder") or hasattr(model, "forward_dispatcher"):
        return model
    if hasattr(model, "module"):
        return find_dit_model(model.module)
    raise ValueError("Cannot find the real model")
