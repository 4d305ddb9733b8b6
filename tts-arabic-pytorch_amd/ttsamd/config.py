"""Model hyper-parameters of the hot path.

`NET_CONFIG` mirrors the reference default (models/fastpitch/__init__.py:3-41) and
`HIFIGAN_CONFIG` the shipped vocoder json (pretrained/hifigan-asc-v1/config.json:1-24);
they are data, restated here because the reference tree does not travel.
"""

NET_CONFIG = {
    'n_mel_channels': 80, 'n_symbols': 148, 'padding_idx': 0,
    'symbols_embedding_dim': 384,
    'in_fft_n_layers': 6, 'in_fft_n_heads': 1, 'in_fft_d_head': 64,
    'in_fft_conv1d_kernel_size': 3, 'in_fft_conv1d_filter_size': 1536,
    'in_fft_output_size': 384,
    'p_in_fft_dropout': 0.1, 'p_in_fft_dropatt': 0.1, 'p_in_fft_dropemb': 0.0,
    'out_fft_n_layers': 6, 'out_fft_n_heads': 1, 'out_fft_d_head': 64,
    'out_fft_conv1d_kernel_size': 3, 'out_fft_conv1d_filter_size': 1536,
    'out_fft_output_size': 384,
    'p_out_fft_dropout': 0.1, 'p_out_fft_dropatt': 0.1, 'p_out_fft_dropemb': 0.0,
    'dur_predictor_kernel_size': 3, 'dur_predictor_filter_size': 256,
    'p_dur_predictor_dropout': 0.1, 'dur_predictor_n_layers': 2,
    'pitch_predictor_kernel_size': 3, 'pitch_predictor_filter_size': 256,
    'p_pitch_predictor_dropout': 0.1, 'pitch_predictor_n_layers': 2,
    'pitch_embedding_kernel_size': 3,
    'n_speakers': 1, 'speaker_emb_weight': 1.0,
    'energy_predictor_kernel_size': 3, 'energy_predictor_filter_size': 256,
    'p_energy_predictor_dropout': 0.1, 'energy_predictor_n_layers': 2,
    'energy_conditioning': True, 'energy_embedding_kernel_size': 3,
}

HIFIGAN_CONFIG = {
    'resblock': '1',
    'upsample_rates': [8, 8, 2, 2],
    'upsample_kernel_sizes': [16, 16, 4, 4],
    'upsample_initial_channel': 512,
    'resblock_kernel_sizes': [3, 7, 11],
    'resblock_dilation_sizes': [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    'num_mels': 80, 'n_fft': 1024, 'hop_size': 256, 'win_size': 1024,
    'sampling_rate': 22050,
}

# vocoder/vocos/__init__.py:35-67 (config_22k): backbone + ISTFT head of MelVocos('22k')
VOCOS_22K_CONFIG = {
    'input_channels': 80, 'dim': 512, 'intermediate_dim': 1536, 'num_layers': 8,
    'n_fft': 1024, 'hop_length': 256, 'padding': 'same',
}

# models/tacotron2/tacotron2_ms.py:152-205 constructor defaults as instantiated by
# models/tacotron2/networks.py:71-82 (n_symbol=40, decoder_max_step=3000)
TACOTRON2_CONFIG = {
    'n_mels': 80, 'n_symbol': 40, 'num_speakers': 40, 'speaker_embedding_dim': 128,
    'symbol_embedding_dim': 512, 'encoder_embedding_dim': 512, 'encoder_n_convolution': 3,
    'encoder_kernel_size': 5, 'decoder_rnn_dim': 1024, 'decoder_max_step': 3000,
    'attention_rnn_dim': 1024, 'attention_hidden_dim': 128, 'attention_location_n_filter': 32,
    'attention_location_kernel_size': 31, 'prenet_dim': 256, 'postnet_n_convolution': 5,
    'postnet_kernel_size': 5, 'postnet_embedding_dim': 512, 'gate_threshold': 0.5, 'decoder_early_stopping': True,
}

SAMPLE_RATE = 22050
HOP = 256

# models/diacritizers/shakkelha/network.py:10-27 and shakkala/network.py:9-24 as tagger geometries
SHAKKELHA_CONFIG = {
    'n_vocab': 91, 'emb_dim': 25, 'lstm_hidden': [256, 256], 'hard_sigmoid': 0, 'bn_after_lstm0': 0, 'bn_eps': 1e-5,
    'dense_dim': [512, 512, 19],
}
SHAKKALA_CONFIG = {
    'n_vocab': 149, 'emb_dim': 288, 'lstm_hidden': [288, 144, 96], 'hard_sigmoid': 1, 'bn_after_lstm0': 1,
    'bn_eps': 1e-3, 'dense_dim': [28],
}
