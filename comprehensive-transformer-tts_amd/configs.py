"""Default hyper-parameter dicts (the values of config/{LJSpeech,VCTK}/*.yaml of the reference).

The reference threads three dicts (preprocess, model, train) through every constructor
(utils/tools.py:19-27); the drop-in consumes the same dicts.  These functions only exist
so that tests / bench.py can build them on a box where the reference tree is absent.
Only keys the hot path reads are included.
"""
import copy
import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

_MODEL = {
    "block_type": "transformer_fs2",
    "duration_modeling": {"learn_alignment": False, "aligner_temperature": 0.0005},
    "prosody_modeling": {
        "model_type": "none",            # "none" | "liu2021"   ("du2021" is outside SURVEY.md section 8)
        "liu2021": {"bottleneck_size_u": 256, "bottleneck_size_p": 4, "ref_enc_filters": [32, 32, 64, 64, 128, 128],
                    "ref_enc_size": [3, 3], "ref_enc_strides": [1, 2], "ref_enc_pad": [1, 1], "ref_enc_gru_size": 32,
                    "ref_attention_dropout": 0.0, "token_num": 32, "predictor_kernel_size": 3, "predictor_dropout": 0.5},
    },
    "transformer_fs2": {
        "encoder_layer": 4, "encoder_head": 2, "encoder_hidden": 256,
        "decoder_layer": 6, "decoder_head": 2, "decoder_hidden": 256,
        "ffn_kernel_size": 9, "encoder_dropout": 0.1, "decoder_dropout": 0.1,
    },
    # VarianceAdaptor / ScheduledOptim read model_config["transformer"]["encoder_hidden"]
    # regardless of block_type (modules.py:739, optimizer.py:20)
    "transformer": {
        "encoder_layer": 4, "encoder_head": 2, "encoder_hidden": 256,
        "decoder_layer": 6, "decoder_head": 2, "decoder_hidden": 256,
        "conv_filter_size": 1024, "conv_kernel_size": [9, 1],
        "encoder_dropout": 0.2, "decoder_dropout": 0.2,
    },
    "conformer": {
        "encoder_layer": 4, "encoder_head": 8, "encoder_hidden": 256,
        "decoder_layer": 6, "decoder_head": 8, "decoder_hidden": 256,
        "feed_forward_expansion_factor": 4, "conv_expansion_factor": 2, "conv_kernel_size": 31,
        "half_step_residual": True, "encoder_dropout": 0.1, "decoder_dropout": 0.1,
    },
    "variance_predictor": {
        "filter_size": 256, "predictor_grad": 0.1, "predictor_layers": 2, "predictor_kernel": 5,
        "cwt_hidden_size": 128, "cwt_std_scale": 0.8, "dur_predictor_layers": 2, "dur_predictor_kernel": 3,
        "dropout": 0.5, "ffn_padding": "SAME", "ffn_act": "gelu",
    },
    "variance_embedding": {
        "use_pitch_embed": True, "pitch_n_bins": 300, "use_energy_embed": True, "energy_n_bins": 256,
        "energy_quantization": "linear",
    },
    "multi_speaker": False,
    "max_seq_len": 1000,
}

_TRAIN = {
    "seed": 1234,
    "optimizer": {
        "batch_size": 16, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "grad_clip_thresh": 1.0,
        "grad_acc_step": 1, "warm_up_step": 4000, "anneal_steps": [300000, 400000, 500000], "anneal_rate": 0.3,
    },
    "loss": {
        "noise_loss": "l1", "dur_loss": "mse", "pitch_loss": "l1", "cwt_loss": "l1", "lambda_f0": 1.0,
        "lambda_uv": 1.0, "lambda_ph_dur": 1.0, "lambda_word_dur": 1.0, "lambda_sent_dur": 1.0,
    },
    "step": {"total_step": 900000, "log_step": 100, "synth_step": 1000, "val_step": 1000, "save_step": 25000,
             "var_start_steps": 50000},
    "duration": {"binarization_start_steps": 6000, "binarization_loss_enable_steps": 18000,
                 "binarization_loss_warmup_steps": 10000},
    "prosody": {"gmm_mdn_beta": 0.02, "prosody_loss_enable_steps": 100000},
}

_PREPROCESS = {
    "dataset": "LJSpeech",
    "path": {"preprocessed_path": os.path.join(_ASSETS, "LJSpeech")},
    "preprocessing": {
        "audio": {"sampling_rate": 22050, "max_wav_value": 32768.0},
        "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
        "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": 8000},
        "pitch": {"pitch_type": "cwt", "pitch_norm": "log", "pitch_norm_eps": 1e-9, "pitch_ar": False,
                  "with_f0": True, "with_f0cwt": True, "use_uv": True, "cwt_scales": list(range(10))},
        "energy": {"feature": "phoneme_level", "normalization": True},
        "speaker_embedder": "none",
    },
}


def get_configs(dataset="LJSpeech"):
    """-> (preprocess_config, model_config, train_config) for the supervised fs2 default."""
    pre, model, train = copy.deepcopy(_PREPROCESS), copy.deepcopy(_MODEL), copy.deepcopy(_TRAIN)
    if dataset == "VCTK":
        pre["dataset"] = "VCTK"
        pre["path"]["preprocessed_path"] = os.path.join(_ASSETS, "VCTK")
        pre["preprocessing"]["speaker_embedder"] = "DeepSpeaker"
        model["multi_speaker"] = True
        model["external_speaker_dim"] = 512
        model["max_seq_len"] = 1500
        train["loss"]["lambda_word_dur"] = 0.0
    elif dataset != "LJSpeech":
        raise ValueError(dataset)
    return pre, model, train


N_SYMBOLS = 360          # len(text.symbols.symbols) in the reference; the embedding has N_SYMBOLS + 1 rows
SIL_PHONEME_IDS = (357, 358, 359)   # text.sil_phonemes_ids(): '@sp', '@spn', '@sil'
