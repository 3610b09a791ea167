"""Model + pipeline glue of the GEN3C denoising path on the HIP kernels.

Mirrors (names, arguments, data-batch keys):
  * `DiffusionGen3CModel` (model/model_gen3c.py:26-139) on top of `DiffusionV2WModel` / `DiffusionT2WModel`
    (model/model_v2w.py:28-259, model/model_t2w.py:124-145): encode/decode through the tokenizer (x sigma_data),
    encode_warped_frames, _get_conditions (cond / uncond incl. zeroed pose for the unconditional branch), the sampling
    loop (delegated to gen3c_amd.sampler.Gen3CDenoiser);
  * the conditioner's two builders for the `video_cond` conditioner (conditioner.py:234-292): text embedding as
    cross-attention context, negative prompt (or zero embeddings) for the unconditional branch;
  * `prepare_data_batch / get_video_batch / create_condition_latent_from_input_frames / compute_num_latent_frames /
    generate_world_from_video` (inference/inference_utils.py:350-455, 542-595, 668-782);
  * `Gen3cPipeline.generate` (inference/gen3c_pipeline.py:108-184) and `_run_tokenizer_decoding`
    (inference/world_generation_pipeline.py:233-247).
T5, MoGe, guardrails and the prompt up-sampler are inputs to / outside of this path: prompts arrive here as
T5 embeddings `[1,512,1024]`.
"""
from __future__ import annotations

import copy
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .dit import VideoExtendGeneralDIT
from .parallel import broadcast, parallel_state
from .sampler import Gen3CDenoiser, VideoExtendCondition, add_condition_video_indicator_and_video_input_mask
from .tokenizer import VideoTokenizer

DEFAULT_AUGMENT_SIGMA = 0.001  # inference_utils.py


def prepare_data_batch(height: int, width: int, num_frames: int, fps: int, prompt_embedding: torch.Tensor,
                       negative_prompt_embedding: Optional[torch.Tensor] = None, device="cuda") -> Dict[str, torch.Tensor]:
    """inference_utils.py:350-406 (the unused uint8 `video` placeholder of the reference is not allocated)."""
    bf = torch.bfloat16
    batch = {
        "t5_text_mask": torch.ones(1, 512, dtype=bf, device=device),
        "image_size": torch.tensor([[height, width, height, width]], dtype=bf, device=device),
        "fps": torch.tensor([fps], dtype=bf, device=device),
        "num_frames": torch.tensor([num_frames], dtype=bf, device=device),
        "padding_mask": torch.zeros((1, 1, height, width), dtype=bf, device=device),
        "t5_text_embeddings": prompt_embedding.to(device=device, dtype=bf),
    }
    if negative_prompt_embedding is not None:
        batch["neg_t5_text_embeddings"] = negative_prompt_embedding.to(device=device, dtype=bf)
        batch["neg_t5_text_mask"] = torch.ones(1, 512, dtype=bf, device=device)
    return batch


class DiffusionGen3CModel:
    def __init__(self, net: VideoExtendGeneralDIT, tokenizer: VideoTokenizer, sigma_data: float = 0.5,
                 latent_shape=(16, 16, 88, 160), frame_buffer_max: int = 2):
        self.net, self.tokenizer = net, tokenizer
        self.sigma_data = sigma_data
        self.state_shape = list(latent_shape)
        self.frame_buffer_max = frame_buffer_max
        self.chunk_size = 121
        self.denoiser = Gen3CDenoiser(net, sigma_data=sigma_data, state_shape=latent_shape)
        self.scheduler = self.denoiser.scheduler
        self.tensor_kwargs = {"device": "cuda", "dtype": torch.bfloat16}

    # ---- tokenizer (model_t2w.py:124-145)
    @torch.no_grad()
    def encode(self, state: torch.Tensor) -> torch.Tensor:
        return self.tokenizer.encode(state) * self.sigma_data

    @torch.no_grad()
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        return self.tokenizer.decode(latent / self.sigma_data)

    # ---- conditions
    def _text_conditions(self, data_batch: dict, is_negative_prompt: bool) -> Tuple[VideoExtendCondition, VideoExtendCondition]:
        """The `video_cond` conditioner's two builders (conditioner.py:234-292, config/base/conditioner.py:27-30):
          * is_negative_prompt=True  = `get_condition_with_negative_prompt`: the TextAttr embedder is NOT dropped for the
            unconditional branch (dropout 0); its input is `neg_t5_text_embeddings` / `neg_t5_text_mask` when the batch carries a
            tensor under that key, otherwise the batch's own (positive) text - this is what Gen3cPipeline always uses
            (gen3c_pipeline.py:250);
          * is_negative_prompt=False = `get_condition_uncondition`: text dropout rate 0.2 > 1e-4 -> rate 1 -> all-zero
            embeddings for the unconditional branch (the mask is never dropped, conditioner.py:102-103)."""
        common = dict(crossattn_mask=data_batch.get("t5_text_mask"), padding_mask=data_batch.get("padding_mask"),
                      fps=data_batch.get("fps"), num_frames=data_batch.get("num_frames"), image_size=data_batch.get("image_size"))
        cond = VideoExtendCondition(crossattn_emb=data_batch["t5_text_embeddings"], **common)
        if is_negative_prompt:
            if isinstance(data_batch.get("neg_t5_text_embeddings"), torch.Tensor):
                un = VideoExtendCondition(crossattn_emb=data_batch["neg_t5_text_embeddings"],
                                          **{**common, "crossattn_mask": data_batch.get("neg_t5_text_mask")})
            else:
                un = VideoExtendCondition(crossattn_emb=data_batch["t5_text_embeddings"], **common)
        else:
            un = VideoExtendCondition(crossattn_emb=torch.zeros_like(data_batch["t5_text_embeddings"]), **common)
        return cond, un

    @torch.no_grad()
    def encode_warped_frames(self, condition_state: torch.Tensor, condition_state_mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """model_gen3c.py:32-57: per cache buffer, tokenizer-encode the render and its (mask*2-1, x3 channels) video;
        zero-pad to frame_buffer_max buffers; concat on channels -> [B, 32*frame_buffer_max, T_lat, h, w]."""
        assert condition_state.dim() == 6
        mask = (condition_state_mask * 2 - 1).repeat(1, 1, 1, 3, 1, 1)
        n_buf = condition_state.shape[2]
        jobs = []
        for i in range(n_buf):
            jobs.append(lambda i=i: self.encode(condition_state[:, :, i].permute(0, 2, 1, 3, 4).to(dtype)).contiguous())
            jobs.append(lambda i=i: self.encode(mask[:, :, i].permute(0, 2, 1, 3, 4).to(dtype)).contiguous())
        # Multi-GPU: the 2N encodes are independent clips - with context parallelism on, clip j is encoded by rank
        # j % cp only and the 7.2 MB latents are all-gathered (the reference encodes all of them on every rank).
        group = self.net.cp_group if self.net.is_context_parallel_enabled else None
        B, T = condition_state.shape[:2]
        tk = self.tokenizer
        shape = (B, tk.channel, tk.get_latent_num_frames(T), condition_state.shape[-2] // tk.spatial_compression_factor,
                 condition_state.shape[-1] // tk.spatial_compression_factor)
        from .parallel import run_jobs_round_robin
        latents = run_jobs_round_robin(jobs, group, shape, dtype, condition_state.device)
        for _ in range(self.frame_buffer_max - n_buf):
            latents += [torch.zeros_like(latents[0]), torch.zeros_like(latents[1])]
        return torch.cat(latents, dim=1)

    def add_condition_pose(self, latent_condition: torch.Tensor, condition: VideoExtendCondition, drop_out_latent: bool = False):
        """model_gen3c.py:115-139."""
        condition.condition_video_pose = torch.zeros_like(latent_condition) if drop_out_latent else latent_condition.contiguous()
        if parallel_state.is_initialized():
            condition.condition_video_pose = broadcast(condition.condition_video_pose, to_tp=True, to_cp=self.net.is_context_parallel_enabled)
        else:
            assert not self.net.is_context_parallel_enabled, "parallel_state is not initialized, context parallel should be turned off."
        return condition

    def _get_conditions(self, data_batch: dict, is_negative_prompt: bool = False, condition_latent: Optional[torch.Tensor] = None,
                        num_condition_t: Optional[int] = None, add_input_frames_guidance: bool = False):
        """model_gen3c.py:59-113."""
        condition, uncondition = self._text_conditions(data_batch, is_negative_prompt)
        latent_condition = self.encode_warped_frames(data_batch["condition_state"], data_batch["condition_state_mask"], self.tensor_kwargs["dtype"])
        condition.video_cond_bool = True
        condition = add_condition_video_indicator_and_video_input_mask(condition_latent, condition, num_condition_t)
        condition = self.add_condition_pose(latent_condition, condition)
        uncondition.video_cond_bool = False if add_input_frames_guidance else True
        uncondition = add_condition_video_indicator_and_video_input_mask(condition_latent, uncondition, num_condition_t)
        uncondition = self.add_condition_pose(latent_condition, uncondition, drop_out_latent=True)
        if parallel_state.is_initialized():  # rank consistency (model_t2w.py:233-240)
            to_cp = self.net.is_context_parallel_enabled
            for c in (condition, uncondition):
                for k, v in c.to_dict().items():
                    if isinstance(v, torch.Tensor):
                        setattr(c, k, broadcast(v, to_tp=False, to_cp=to_cp))
        return condition, uncondition

    @torch.no_grad()
    def generate_samples_from_batch(self, data_batch: dict, guidance: float = 1.5, seed: int = 1, state_shape=None, n_sample: int = 1,
                                    is_negative_prompt: bool = False, num_steps: int = 35, condition_latent: Optional[torch.Tensor] = None,
                                    num_condition_t: Optional[int] = None, condition_augment_sigma: float = None,
                                    add_input_frames_guidance: bool = False, xt: Optional[torch.Tensor] = None) -> torch.Tensor:
        """model_v2w.py:84-155."""
        assert condition_latent is not None, "condition_latent should be provided"
        condition, uncondition = self._get_conditions(data_batch, is_negative_prompt, condition_latent, num_condition_t, add_input_frames_guidance)
        return self.denoiser.generate_samples_from_batch(condition, uncondition, guidance=guidance, seed=seed,
                                                         state_shape=state_shape or self.state_shape, n_sample=n_sample, num_steps=num_steps,
                                                         condition_augment_sigma=condition_augment_sigma, xt=xt)


def compute_num_latent_frames(model: DiffusionGen3CModel, num_input_frames: int, downsample_factor: int = 8) -> int:
    """inference_utils.py:668-692."""
    tk = model.tokenizer
    n = num_input_frames // tk.pixel_chunk_duration * tk.latent_chunk_duration
    if num_input_frames % tk.latent_chunk_duration == 1:
        n += 1
    elif num_input_frames % tk.latent_chunk_duration > 1:
        assert (num_input_frames % tk.pixel_chunk_duration - 1) % downsample_factor == 0
        n += 1 + (num_input_frames % tk.pixel_chunk_duration - 1) // downsample_factor
    return n


def create_condition_latent_from_input_frames(model: DiffusionGen3CModel, input_frames: torch.Tensor, num_frames_condition: int = 1):
    """inference_utils.py:695-760 (condition_location 'first_n'): last `num_frames_condition` frames, zero-padded to one
    pixel chunk, encoded."""
    B, C, T, H, W = input_frames.shape
    n_enc = model.tokenizer.pixel_chunk_duration
    assert T >= num_frames_condition and n_enc >= num_frames_condition
    cond = input_frames[:, :, -num_frames_condition:]
    enc_in = torch.cat([cond, cond.new_zeros(B, C, n_enc - num_frames_condition, H, W)], dim=2)
    return model.encode(enc_in), enc_in


def generate_world_from_video(model: DiffusionGen3CModel, state_shape, is_negative_prompt: bool, data_batch: dict, guidance: float,
                              num_steps: int, seed: int, condition_latent: torch.Tensor, num_input_frames: int,
                              xt: Optional[torch.Tensor] = None) -> torch.Tensor:
    """inference_utils.py:542-595."""
    if condition_latent.shape[2] < state_shape[1]:
        b, c, t, h, w = condition_latent.shape
        condition_latent = torch.cat([condition_latent, condition_latent.new_zeros(b, c, state_shape[1] - t, h, w)], dim=2).contiguous()
    return model.generate_samples_from_batch(data_batch, guidance=guidance, state_shape=state_shape, num_steps=num_steps,
                                             is_negative_prompt=is_negative_prompt, seed=seed, condition_latent=condition_latent,
                                             num_condition_t=compute_num_latent_frames(model, num_input_frames),
                                             condition_augment_sigma=DEFAULT_AUGMENT_SIGMA, xt=xt)


class Gen3cPipeline:
    """gen3c_pipeline.py:36-184 without the model-loading / offloading / guardrail / text-encoder plumbing: takes ready
    T5 embeddings and the rendered 3D-cache buffers, returns the uint8 video."""

    def __init__(self, model: DiffusionGen3CModel, guidance: float = 1.0, num_steps: int = 35, height: int = 704, width: int = 1280,
                 fps: int = 24, num_video_frames: int = 121, seed: int = 1, text_encoder=None):
        """text_encoder: callable str -> T5 embedding [1,512,1024] (cli_common.TextEmbedder; T5-11B itself is an input of the path).
        Only generate() with string prompts needs it."""
        self.model, self.guidance, self.num_steps = model, guidance, num_steps
        self.height, self.width, self.fps, self.num_video_frames, self.seed = height, width, fps, num_video_frames, seed
        self.text_encoder = text_encoder

    @torch.no_grad()
    def generate(self, prompt, image_path, rendered_warp_images: torch.Tensor, rendered_warp_masks: torch.Tensor,
                 negative_prompt=None, xt: Optional[torch.Tensor] = None):
        """The reference's Gen3cPipeline.generate seam (gen3c_pipeline.py:108-184; called with exactly these keywords by
        gen3c_single_image.py:366-372, 411-417): returns (uint8 video [T,H,W,3], prompt).
          prompt / negative_prompt: str (embedded by `text_encoder`, world_generation_pipeline.py:188-231) or a ready T5 embedding tensor;
          image_path: path of the conditioning image, or - as the reference's autoregressive loop passes it - a [1,3,1,H,W] tensor in [-1,1].
        The guardrail / prompt-upsampler steps of the reference are control plane and not part of this path."""
        def embed(p_):
            if p_ is None or isinstance(p_, torch.Tensor):
                return p_
            if self.text_encoder is None:
                raise RuntimeError("Gen3cPipeline.generate got a string prompt but no text_encoder: construct the pipeline with "
                                   "text_encoder=cli_common.TextEmbedder(...) or pass the T5 embedding tensor")
            return self.text_encoder(p_)
        if isinstance(image_path, str):
            from .gen3c_single_image import load_condition_image
            image = load_condition_image(image_path, self.height, self.width)
        else:
            image = image_path
        # a ready embedding TENSOR has no truth value ("Boolean value of Tensor with more than one value is ambiguous"); "" / None = no negative prompt
        has_negative = negative_prompt is not None and not (isinstance(negative_prompt, str) and not negative_prompt)
        video = self.generate_from_embeddings(embed(prompt), image, rendered_warp_images, rendered_warp_masks,
                                              negative_prompt_embedding=embed(negative_prompt) if has_negative else None, xt=xt)
        return video, prompt

    @torch.no_grad()
    def generate_from_embeddings(self, prompt_embedding: torch.Tensor, image: torch.Tensor, rendered_warp_images: torch.Tensor,
                                 rendered_warp_masks: torch.Tensor, negative_prompt_embedding: Optional[torch.Tensor] = None,
                                 xt: Optional[torch.Tensor] = None) -> np.ndarray:
        """image: [1,3,1,H,W] in [-1,1] (the reference reads uint8/128-1, inference_utils.py:648);
        rendered_warp_images [1,121,N,3,H,W], rendered_warp_masks [1,121,N,1,H,W] (Cache3D.render_cache).
        -> uint8 [121,H,W,3]."""
        dev = self.model.net.affline_norm.weight.device
        condition_latent, _ = create_condition_latent_from_input_frames(self.model, image.to(dev, torch.bfloat16), 1)
        condition_latent = condition_latent.to(torch.bfloat16)
        batch = prepare_data_batch(self.height, self.width, self.num_video_frames, self.fps, prompt_embedding, negative_prompt_embedding, dev)
        batch["condition_state"] = rendered_warp_images.to(dev)
        batch["condition_state_mask"] = rendered_warp_masks.to(dev)
        # is_negative_prompt=True unconditionally, as Gen3cPipeline._run_model does (gen3c_pipeline.py:247-251)
        sample = generate_world_from_video(self.model, self.model.state_shape, True, batch, self.guidance,
                                           self.num_steps, self.seed, condition_latent, 1, xt=xt)
        video = (1.0 + self.model.decode(sample)).clamp(0, 2) / 2  # world_generation_pipeline.py:244-245
        return (video[0].permute(1, 2, 3, 0) * 255).to(torch.uint8).cpu().numpy()
