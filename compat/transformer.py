"""`from transformer import Network` (styletransformer/seq2seq.py:16) -> nast_amd.transformer, every public name of
styletransformer/transformer.py."""
from nast_amd.transformer import (Network, StyleTransformer, Discriminator, Encoder, Decoder, Generator, EmbeddingLayer,  # noqa: F401
                                  EncoderLayer, DecoderLayer, MultiHeadAttention, scaled_attention, PositionwiseFeedForward,
                                  SublayerConnection, Linear, Embedding, LayerNorm, HiddenDecoder, LengthPredictor)
