"""MI355X-native FastConformer encoder behind the reference's module API.

    from conformer_amd import create_model, load_config
    model = create_model(load_config("fastconformer_small.json")).cuda().eval()
    out, out_len, _ = model.encode(mel, mel_lengths)        # runs libfcenc_hip.so
"""
from .model import (FastConformerTransducer, JointNetwork, StatelessPredictor, StochasticDepthConfig,
                    create_model, load_config, model_size)
from .frontend import MelSpectrogram
from .modules import (CausalConv1d, ConformerConvolution, FastConformerBlock, FastConformerEncoder,
                      FeedForwardModule, RelativePositionalEncoding, RelPositionMultiHeadAttention, Subsampling8x,
                      stochastic_depth_schedule)

__all__ = ["create_model", "load_config", "model_size", "FastConformerTransducer", "FastConformerEncoder",
           "FastConformerBlock", "Subsampling8x", "RelativePositionalEncoding", "RelPositionMultiHeadAttention",
           "ConformerConvolution", "FeedForwardModule", "CausalConv1d", "StatelessPredictor", "JointNetwork",
           "StochasticDepthConfig", "stochastic_depth_schedule", "MelSpectrogram"]
