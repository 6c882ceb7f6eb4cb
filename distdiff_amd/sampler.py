"""Python shims with the reference's sampler signatures (generate_data.py:109, 687-689, 735-737) over the C ABI.

A maintainer of the reference can swap these in for the three functions: `unet` / `vae` / `image_encoder` /
`noise_scheduler` are replaced by one `Engine` (passed as `unet`); the other model arguments are accepted and ignored.
Random draws follow the reference: e ~ U[0,1) and b ~ N(0,1) from the CPU global generator (:692-695)."""
import torch


def _step_index(engine, t):
    return engine.timesteps.index(int(t))


def _prototypes(engine, total_global_proto, total_local_proto):
    """The reference passes the prototype tables with every call (:1204-1213); the engine keeps them on the device, so they are
    uploaded when the caller hands over different tensors than last time (None, None = keep what dd_set_prototypes installed)."""
    if total_global_proto is None and total_local_proto is None:
        return
    # the cache holds the tensors themselves (an id() alone can be recycled by a new tensor at the same address) and their in-place
    # modification counters, so a table that was freed, replaced or updated in place is uploaded again
    last = getattr(engine, "_shim_protos", None)
    cur = (total_global_proto, total_local_proto)
    ver = tuple(getattr(t, "_version", None) for t in cur)
    if last is None or last[0][0] is not cur[0] or last[0][1] is not cur[1] or last[1] != ver:
        engine.set_prototypes(total_global_proto, total_local_proto)
        engine._shim_protos = (cur, ver)


def denoise_one_step(latents, noise_scheduler, t, unet, prompt_embeds, class_labels):
    """-> (latents, x_0). `unet` is the Engine; prompt_embeds = cat[negative, prompt] is installed with engine.set_prompt."""
    engine = unet
    if prompt_embeds is not None:
        engine.set_prompt(prompt_embeds)
    return engine.denoise_step(latents, _step_index(engine, t))


def transform_guidance(latents, batch, sub_timesteps, noise_scheduler, unet, prompt_embeds, class_labels, vae, image_encoder,
                       image_processor, weight_dtype, generator, total_global_proto, total_local_proto):
    """-> (latents, score)."""
    engine = unet
    bs, ch = latents.shape[0], latents.shape[1]
    channel_noise = torch.rand([bs, ch, 1, 1])                                  # :692
    channel_noise_bias = torch.zeros([bs, ch, 1, 1]).normal_(0, 1)              # :694
    if prompt_embeds is not None:
        engine.set_prompt(prompt_embeds)
    _prototypes(engine, total_global_proto, total_local_proto)
    first = _step_index(engine, sub_timesteps[0])
    z, score, _ = engine.transform_guidance(latents, batch["targets"], channel_noise, channel_noise_bias, first, len(sub_timesteps))
    return z, score[0]


def direct_guidance(latents, batch, t_i, noise_scheduler, unet, prompt_embeds, class_labels, vae, image_encoder, image_processor,
                    weight_dtype, generator, total_global_proto, total_local_proto):
    """-> (latents, x_0, score)."""
    engine = unet
    if prompt_embeds is not None:
        engine.set_prompt(prompt_embeds)
    _prototypes(engine, total_global_proto, total_local_proto)
    zn, x0, score, _ = engine.direct_guidance(latents, batch["targets"], _step_index(engine, t_i))
    return zn, x0, score[0]
