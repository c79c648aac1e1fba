"""DDIM scheduler (host-side coefficients; the tensor update runs in the fused cfg+ddim HIP kernel).

Drop-in for diffusers.DDIMScheduler as configured by configs/inference/inference_v2.yaml:24-33 and used at
src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:373,182,519-521,551-553."""
import numpy as np
import torch


def _rescale_zero_terminal_snr(betas):
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    s = alphas_cumprod.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * s0 / (s0 - sT)
    abar = s ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


class _StepOut:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 timestep_spacing="leading", rescale_betas_zero_snr=False, **unused):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            betas = _rescale_zero_terminal_snr(betas)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        if prediction_type != "v_prediction" or clip_sample:
            raise NotImplementedError("the fused HIP step implements v_prediction without sample clipping "
                                      "(the reference's inference_v2.yaml settings)")
        self.num_train_timesteps, self.steps_offset, self.timestep_spacing = num_train_timesteps, steps_offset, timestep_spacing
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64) + self.steps_offset
        elif self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)  # kept on the host: they only parameterise kernel launches

    def coefficients(self, timestep):
        """(sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev)) as python floats (fp32 arithmetic like diffusers)."""
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, **unused):
        """Tensor-level API kept for drop-in compatibility (runs the same HIP kernel with guidance off)."""
        from . import ops
        assert eta == 0.0
        lat = sample.detach().float().contiguous().clone()
        acc = model_output.detach().float().contiguous()
        cnt = torch.ones((lat.shape[2],), device=lat.device)
        ops.cfg_ddim_step(acc, cnt, lat, False, 1.0, *self.coefficients(timestep))
        return _StepOut(lat.to(sample.dtype))
