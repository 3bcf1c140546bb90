"""TEST INFRASTRUCTURE - CPU restatement of the reference's loss (model/loss.py:10-386) in stock torch ops.

Checker only (tests/, bench.py's cpu_baseline): the product loss (comprehensive-transformer-tts_amd/loss.py) runs on the HIP device and
raises on host tensors.  Pinned against the reference's own 9-tuple on goldens G9 / G6-loss / G10-loss (tests/test_loss_cpu.py).
Two formulation changes that do not alter any value: the word-duration scatter uses the static bound Ts+1 instead of `word_id.max()+1`
(loss.py:156-157) and the energy L1 is a masked mean instead of `masked_select` (loss.py:236-243); ForwardSumLoss is one batched
`F.ctc_loss` call with the classes beyond each utterance's key length masked to -inf (the reference loops over utterances).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

SIL_PHONEME_IDS = (357, 358, 359)        # "@sp", "@spn", "@sil" of text/symbols.py (the silence tokens loss.py:30-33 looks up)


class RefLoss(nn.Module):
    def __init__(self, preprocess_config, model_config, train_config):
        super().__init__()
        self.learn_alignment = model_config["duration_modeling"]["learn_alignment"]
        self.binarization_loss_enable_steps = train_config["duration"]["binarization_loss_enable_steps"]
        self.binarization_loss_warmup_steps = train_config["duration"]["binarization_loss_warmup_steps"]
        self.loss_config = train_config["loss"]
        self.pitch_config = preprocess_config["preprocessing"]["pitch"]
        self.pitch_type = self.pitch_config["pitch_type"]
        self.energy_feature_level = preprocess_config["preprocessing"]["energy"]["feature"]
        self.use_pitch_embed = model_config["variance_embedding"]["use_pitch_embed"]
        self.use_energy_embed = model_config["variance_embedding"]["use_energy_embed"]
        self.var_start_steps = train_config["step"]["var_start_steps"]
        self.model_type = model_config["prosody_modeling"]["model_type"]
        self.prosody_loss_enable_steps = train_config["prosody"]["prosody_loss_enable_steps"]
        self.sil_ph_ids = SIL_PHONEME_IDS

    @staticmethod
    def _masked_l1_mel(pred, target, pad_mask):
        pred = pred.masked_fill(pad_mask.unsqueeze(-1), 0)
        target = target.masked_fill(pad_mask.unsqueeze(-1), 0)
        w = target.abs().sum(-1, keepdim=True).ne(0).float().expand_as(target)
        return ((pred - target).abs() * w).sum() / w.sum()

    def _duration_loss(self, dur_pred, dur_gt, txt_tokens, nonpad):
        losses = {}
        B, T = txt_tokens.shape
        dur_gt = dur_gt.float() * nonpad
        is_sil = torch.zeros_like(txt_tokens).bool()
        for p_id in self.sil_ph_ids:
            is_sil = is_sil | (txt_tokens == p_id)
        is_sil = is_sil.float()
        pd = F.mse_loss(dur_pred, (dur_gt + 1).log(), reduction="none")
        losses["pdur"] = (pd * nonpad).sum() / nonpad.sum() * self.loss_config["lambda_ph_dur"]
        dur_lin = (dur_pred.exp() - 1).clamp(min=0)
        if self.loss_config["lambda_word_dur"] > 0:
            word_id = (is_sil.cumsum(-1) * (1 - is_sil)).long()
            wp = dur_lin.new_zeros([B, T + 1]).scatter_add(1, word_id, dur_lin)[:, 1:]
            wg = dur_gt.new_zeros([B, T + 1]).scatter_add(1, word_id, dur_gt)[:, 1:]
            wl = F.mse_loss((wp + 1).log(), (wg + 1).log(), reduction="none")
            wn = (wg > 0).float()
            losses["wdur"] = (wl * wn).sum() / wn.sum() * self.loss_config["lambda_word_dur"]
        if self.loss_config["lambda_sent_dur"] > 0:
            sl = F.mse_loss((dur_lin.sum(-1) + 1).log(), (dur_gt.sum(-1) + 1).log(), reduction="mean")
            losses["sdur"] = sl.mean() * self.loss_config["lambda_sent_dur"]
        return losses

    def _pitch_loss(self, p_pred, p_tgt, mel_nonpad, src_nonpad=None):
        lam = self.loss_config["lambda_f0"]
        losses = {}
        if self.pitch_type != "cwt":
            fn = F.l1_loss if self.loss_config["pitch_loss"] == "l1" else F.mse_loss          # loss.py:175,217 ('ssim' raises there)
            pred = p_pred["pitch_pred"]
            if self.pitch_type == "ph":                                                       # loss.py:173-178
                losses["f0"] = (fn(pred[:, :, 0], p_tgt["f0"], reduction="none") * src_nonpad).sum() / src_nonpad.sum() * lam
                return losses
            nonpad = mel_nonpad                                                               # loss.py:202-219 add_f0_loss
            if self.pitch_config["use_uv"]:
                losses["uv"] = ((F.binary_cross_entropy_with_logits(pred[:, :, 1], p_tgt["uv"], reduction="none") * nonpad).sum()
                                / nonpad.sum() * self.loss_config["lambda_uv"])
                nonpad = nonpad * (p_tgt["uv"] == 0).float()
            losses["f0"] = (fn(pred[:, :, 0], p_tgt["f0"], reduction="none") * nonpad).sum() / nonpad.sum() * lam
            return losses
        cwt_pred = p_pred["cwt"][:, :, :10]
        if self.loss_config["cwt_loss"] == "l1":
            losses["C"] = F.l1_loss(cwt_pred, p_tgt["cwt_spec"]) * lam
        else:
            losses["C"] = F.mse_loss(cwt_pred, p_tgt["cwt_spec"]) * lam
        if self.pitch_config["use_uv"]:
            uv_pred = p_pred["cwt"][:, :, -1]
            losses["uv"] = ((F.binary_cross_entropy_with_logits(uv_pred, p_tgt["uv"], reduction="none") * mel_nonpad).sum()
                            / mel_nonpad.sum() * self.loss_config["lambda_uv"])
        losses["f0_mean"] = F.l1_loss(p_pred["f0_mean"], p_tgt["f0_mean"]) * lam
        losses["f0_std"] = F.l1_loss(p_pred["f0_std"], p_tgt["f0_std"]) * lam
        return losses

    @staticmethod
    def forward_sum_loss(attn_logprob, in_lens, out_lens, blank_logprob=-1.0, host_lens=None):
        """ForwardSumLoss (loss.py:350-377) as ONE batched CTC call instead of a per-sample Python loop: classes beyond
        key_len are excluded from each sample's log-softmax (the reference slices them away) by masking them to -inf.
        `host_lens=(in_list, out_list)`: the same lengths as Python ints - F.ctc_loss otherwise copies the device tensors
        to the host (a sync that a hipGraph capture cannot contain)."""
        B, _, Tm, Ts = attn_logprob.shape
        logits = F.pad(attn_logprob[:, 0], (1, 0), value=blank_logprob)                     # [B,Tm,Ts+1], class 0 = blank
        cls = torch.arange(Ts + 1, device=logits.device)[None, None, :]
        logits = logits.masked_fill(cls > in_lens[:, None, None], float("-inf"))
        logp = torch.log_softmax(logits, dim=-1).transpose(0, 1)                             # [Tm,B,Ts+1]
        targets = torch.arange(1, Ts + 1, device=logits.device)[None, :].expand(B, -1)
        ctc_in, ctc_tgt = (list(host_lens[1]), list(host_lens[0])) if host_lens is not None else (out_lens, in_lens)
        per = F.ctc_loss(logp, targets, ctc_in, ctc_tgt, blank=0, reduction="none", zero_infinity=True)
        return (per / in_lens.clamp(min=1).to(per.dtype)).sum() / B          # nn.CTCLoss 'mean' per sample, then / batch

    @staticmethod
    def bin_loss(hard, soft):
        """BinLoss (loss.py:380-386) with a mask product instead of boolean indexing (no host sync)."""
        return -(torch.log(torch.clamp(soft, min=1e-12)) * hard).sum() / hard.sum()

    def forward(self, inputs, predictions, step):
        (texts, _, _, mel_targets, _, _, pitch_targets, energy_targets, duration_targets, _, _) = inputs[3:]
        (mel_pred, post_pred, p_pred, e_pred, log_d, _, src_masks, mel_masks, src_lens, mel_lens, attn_outs, prosody_info) = predictions
        src_nonpad = (~src_masks)
        mel_nonpad = (~mel_masks)
        mel_targets = mel_targets[:, : mel_masks.shape[1], :]
        mel_loss = self._masked_l1_mel(mel_pred, mel_targets, mel_masks)
        postnet_mel_loss = self._masked_l1_mel(post_pred, mel_targets, mel_masks)
        zero = torch.zeros(1, device=mel_targets.device)
        ctc_loss = bin_loss = zero
        if self.learn_alignment:
            attn_soft, attn_hard, attn_hard_dur, attn_logprob = attn_outs
            duration_targets = attn_hard_dur
            ctc_loss = self.forward_sum_loss(attn_logprob, src_lens, mel_lens, host_lens=None)
            if step < self.binarization_loss_enable_steps:
                w = 0.0
            else:
                w = min((step - self.binarization_loss_enable_steps) / self.binarization_loss_warmup_steps, 1.0)
            bin_loss = self.bin_loss(attn_hard, attn_soft) * w
        prosody_loss = zero
        if self.training and self.model_type == "liu2021" and step > self.prosody_loss_enable_steps:
            # loss.py:319-324.  The phoneme-level term selects with `src_masks` (True = PAD) exactly as the reference does:
            # it averages |pp_tgt - pp_vec| over the PADDED phoneme positions (0/0 = NaN for a batch without padding).
            up_tgt, pp_tgt, up_vec, pp_vec, _ = prosody_info
            sel = src_masks.unsqueeze(-1).to(pp_vec.dtype)
            prosody_loss = F.l1_loss(up_tgt, up_vec) + ((pp_tgt - pp_vec).abs() * sel).sum() / (sel.sum() * pp_vec.shape[-1])
        total = mel_loss + postnet_mel_loss + ctc_loss + bin_loss + prosody_loss + zero
        duration_loss = {"pdur": zero, "wdur": zero, "sdur": zero}
        if self.pitch_type == "cwt":                  # get_init_losses, loss.py:241-264
            pitch_loss = {"C": zero, "uv": zero, "f0_mean": zero, "f0_std": zero}
        elif self.pitch_type == "ph":
            pitch_loss = {"f0": zero}
        else:
            pitch_loss = dict(**({"uv": zero} if self.pitch_config["use_uv"] else {}), f0=zero)
        energy_loss = zero
        if step > self.var_start_steps:
            duration_loss = self._duration_loss(log_d, duration_targets, texts, src_nonpad.float())
            if self.use_pitch_embed:
                pitch_loss = self._pitch_loss(p_pred, pitch_targets, mel_nonpad.float(), src_nonpad.float())
            if self.use_energy_embed:
                m = (src_nonpad if self.energy_feature_level == "phoneme_level" else mel_nonpad).float()      # loss.py:236-241
                energy_loss = ((e_pred - energy_targets).abs() * m).sum() / m.sum()
            total = total + sum(duration_loss.values()) + sum(pitch_loss.values()) + energy_loss
        return (total, mel_loss, postnet_mel_loss, pitch_loss, energy_loss, duration_loss, ctc_loss, bin_loss, prosody_loss)
