"""echo-tts_amd — MI355X-native hot path of Echo-TTS (sampler + EchoDiT + Fish S1-DAC decode, and the
DAC encode of the speaker reference).

Importable as `echo_tts_amd` (the directory name carries a hyphen).  Importing does not touch
the GPU; any compute entry point raises `EchoHipError` when libechohip.so or a gfx950 device is
missing — there is deliberately no CPU fallback.
"""
from ._lib import EchoHipError, load_library  # noqa: F401
from .model import EchoDiT, EchoDiTConfig  # noqa: F401
from .autoencoder import DAC, DACConfig  # noqa: F401
from .inference import (  # noqa: F401
    PCAState, ae_decode, ae_encode, ae_reconstruct, chunk_text, crop_audio_to_flattening_point, find_flattening_point,
    get_speaker_latent_and_mask, get_text_input_ids_and_mask, sample_euler_cfg_independent_guidances, sample_pipeline,
    sample_pipeline_chunked, tokenizer_encode,
)
from .inference_blockwise import sample_blockwise, sample_blockwise_euler_cfg_independent_guidances  # noqa: F401
from .audio_io import load_audio, read_wav, resample, wav_bytes  # noqa: F401
