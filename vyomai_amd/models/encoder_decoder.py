"""BART-style encoder-decoder with the reference's API (VyomAI/models/encoder_decoder.py).

Decoder layer = self-attention -> cross-attention over the encoder output -> FeedForward whose
residual is the LAYER INPUT (reference :57-83).  Cross-attention keys/values are projected once per
generation when a cache is attached (layers/attention.py:440-462).  The encoder padding mask is a
key-padding descriptor evaluated inside the attention kernels, not a (B,1,1,S) additive tensor.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU_ERF
from ..layers.attention import (DecoderAttention, DecoderAttentionGqa, EncoderDecoderAttention,
                                EncoderDecoderAttentionGqa, _shadow)
from ..layers.ffn import FeedForward
from ..layers.kv_cache import DynamicCache, StaticCache  # noqa: F401  (re-exported like the reference)
from ..layers.mask import AttnMask
from ..autograd_train import defer_residual_grads as _defer
from .common import PositionMixin
from .encoder import EncoderModel


@dataclass
class Seq2SeqOutput(object):
    logits: torch.Tensor
    key_value_states: torch.Tensor


class Seq2SeqDecoderLayer(nn.Module):
    """decoder layer for Seq2Seq model.  Reference :33-83."""

    def __init__(self, config, layer_idx: Optional[int] = 0, attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.attention = (DecoderAttentionGqa(config, layer_idx=layer_idx) if attention_type == "gqa"
                          else DecoderAttention(config, layer_idx=layer_idx))
        if attention_type == "gqa" and layer_idx == 0:
            print("Decoder Using GQA Attention")
        self.cross_attention = (EncoderDecoderAttentionGqa(config, layer_idx=layer_idx) if attention_type == "gqa"
                                else EncoderDecoderAttention(config, layer_idx=layer_idx))
        if attention_type == "gqa" and layer_idx == 0:
            print("Using GQA in Cross Attention")
        self.feed_forward = FeedForward(config)
        self.layer_idx = layer_idx

    def forward(self, hidden_state: torch.Tensor, attention_mask, encoder_hidden_state: Optional[torch.Tensor] = None,
                encoder_attention_mask=None, freqs=None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> torch.Tensor:
        _defer(hidden_state)  # training: its residual-path gradients are added in the QKV dgrad epilogue
        out = self.attention(hidden_state=hidden_state, attention_mask=attention_mask, freqs=freqs,
                             use_cache=use_cache, start_pos=start_pos)
        out = self.cross_attention(hidden_state=out, encoder_hidden_state=encoder_hidden_state,
                                   encoder_attention_mask=encoder_attention_mask, freqs=freqs, use_cache=use_cache)
        return self.feed_forward(out, hidden_state)


class LMHead(nn.Module):
    """vocab(LN(gelu(dense(h)))) with the vocabulary bias tied to ``vocab.bias`` (reference :86-110;
    note the projection is called ``vocab`` here, ``decoder`` in models/decoder.py)."""

    def __init__(self, config) -> None:
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-6))
        self.vocab = nn.Linear(config.hidden_size, config.vocab_size)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.vocab.bias = self.bias

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        from ..autograd import _wants_grad
        if _wants_grad(hidden_state, self.dense.weight, self.vocab.weight):
            from ..autograd_train import LMHeadFn
            return LMHeadFn.apply(hidden_state, self.dense.weight, self.dense.bias, self.layer_norm.weight,
                                  self.layer_norm.bias, self.vocab.weight, self.bias, self.layer_norm.eps)
        dt = hidden_state.dtype
        x = ops.linear(hidden_state, _shadow(self.dense.weight, dt), _shadow(self.dense.bias, dt), act=ACT_GELU_ERF)
        x, _, _ = ops.layernorm(x, _shadow(self.layer_norm.weight, dt), _shadow(self.layer_norm.bias, dt),
                                self.layer_norm.eps)
        return ops.linear(x, _shadow(self.vocab.weight, dt), _shadow(self.bias, dt))

    def loss(self, hidden_state: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
        """Shifted next-token cross-entropy fused with the head (see autograd_train.LMHeadLossFn)."""
        from ..autograd_train import LMHeadLossFn
        return LMHeadLossFn.apply(hidden_state, labels, ignore_index, self.dense.weight, self.dense.bias,
                                  self.layer_norm.weight, self.layer_norm.bias, self.vocab.weight, self.bias,
                                  self.layer_norm.eps)


class Seq2SeqDecoderModel(nn.Module, PositionMixin):
    """Seq2Seq decoder model.  Reference :113-258."""

    def __init__(self, config, pos_embedding_type: Optional[str] = "absolute",
                 attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                            padding_idx=getattr(config, "pad_token_id", None))
        self._init_positions(config, pos_embedding_type, "Decoder")
        self.all_layer = nn.ModuleList(
            [Seq2SeqDecoderLayer(config, i, attention_type=attention_type) for i in range(config.num_hidden_layers)])

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                encoder_hidden_state: Optional[torch.Tensor] = None, encoder_attention_mask=None,
                use_cache: Optional[bool] = False, start_pos: Optional[int] = 0) -> torch.Tensor:
        _bsz, seqlen = input_ids.shape
        hidden_state = self._embed(self.word_embeddings, input_ids)
        hidden_state, freqs = self._positions(hidden_state, start_pos, seqlen)
        mask = None
        if seqlen > 1:
            mask = self.create_mask_for_decoder(input_ids=input_ids, attention_mask=attention_mask,
                                                start_pos=start_pos)
        if encoder_hidden_state is not None and encoder_hidden_state.dtype != hidden_state.dtype:
            encoder_hidden_state = encoder_hidden_state.to(hidden_state.dtype)
        for layer in self.all_layer:
            hidden_state = layer(hidden_state=hidden_state, attention_mask=mask,
                                 encoder_hidden_state=encoder_hidden_state,
                                 encoder_attention_mask=encoder_attention_mask, freqs=freqs, use_cache=use_cache,
                                 start_pos=start_pos)
        return hidden_state

    def create_mask_for_decoder(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                                start_pos: Optional[int] = 0) -> AttnMask:
        """The reference (:214-254) returns the dense (B,1,L,start+L) 0/1 product of the causal and
        padding masks; here the same information as a descriptor (``.dense()`` rebuilds the tensor)."""
        _, seq_length = input_ids.shape
        return AttnMask.from_padding(attention_mask, causal=True, start_pos=start_pos, query_len=seq_length)

    @classmethod
    def from_config(cls, config) -> nn.Module:
        return cls(config)


class EncoderDecoderModel(nn.Module):
    """Reference :261-391."""

    def __init__(self, encoder_config, decoder_config, encoder: Optional[nn.Module] = None,
                 encoder_pos_embedding_type: Optional[str] = "absolute", encoder_attention_type: Optional[str] = None,
                 decoder_pos_embedding_type: Optional[str] = "absolute",
                 decoder_attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.is_gqa = True if decoder_attention_type == "gqa" else False
        self.encoder = (encoder if encoder is not None
                        else EncoderModel(config=encoder_config, pos_embedding_type=encoder_pos_embedding_type,
                                          attention_type=encoder_attention_type))
        self.decoder = Seq2SeqDecoderModel(config=decoder_config, pos_embedding_type=decoder_pos_embedding_type,
                                           attention_type=decoder_attention_type)
        self.lm_head = LMHead(config=decoder_config)

    def _decoder_hidden(self, input_ids, attention_mask, decoder_input_ids, decoder_attention_mask, encoder_output,
                        use_cache, start_pos):
        if encoder_output is None:
            encoder_output = self.encoder(input_ids=input_ids, attention_mask=attention_mask).logits
        # reference :319-331: a missing mask means all ones; (1-mask)*finfo.min of shape (B,1,1,S) is
        # the same information as a key-padding descriptor (None when nothing is padded)
        enc_mask = None
        if attention_mask is not None:
            enc_mask = AttnMask.from_padding(attention_mask, causal=False, start_pos=0,
                                             query_len=decoder_input_ids.shape[1])
            if enc_mask.keypad is None:
                enc_mask = None
        hidden = self.decoder(input_ids=decoder_input_ids, attention_mask=decoder_attention_mask,
                              encoder_hidden_state=encoder_output, encoder_attention_mask=enc_mask,
                              use_cache=use_cache, start_pos=start_pos)
        return hidden, encoder_output

    def forward(self, input_ids: Optional[torch.LongTensor] = None, attention_mask: Optional[torch.Tensor] = None,
                decoder_input_ids: Optional[torch.LongTensor] = None,
                decoder_attention_mask: Optional[torch.LongTensor] = None,
                encoder_output: Optional[torch.FloatTensor] = None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> Seq2SeqOutput:
        hidden, encoder_output = self._decoder_hidden(input_ids, attention_mask, decoder_input_ids,
                                                      decoder_attention_mask, encoder_output, use_cache, start_pos)
        return Seq2SeqOutput(key_value_states=encoder_output, logits=self.lm_head(hidden))

    def seq2seq_loss(self, input_ids, attention_mask, decoder_input_ids, labels, decoder_attention_mask=None,
                     ignore_index: int = -100) -> torch.Tensor:
        """Shifted next-token loss of the decoder with the LM head and cross-entropy fused (training
        entry point; the reference trains this model with the notebook CE on `logits`)."""
        hidden, _ = self._decoder_hidden(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask,
                                         None, False, 0)
        return self.lm_head.loss(hidden, labels, ignore_index)

    def get_encoder(self) -> nn.Module:
        return self.encoder

    def get_encoder_output(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> object:
        return self.encoder(input_ids=input_ids, attention_mask=attention_mask)

    def get_decoder(self) -> Seq2SeqDecoderModel:
        return self.decoder

    def _setup_cache(self, config, cls: Optional[object] = StaticCache) -> None:
        """kv-cache hooks for every self-attention and cross-attention layer (reference :354-358)."""
        for layer in self.decoder.all_layer:
            layer.attention.cache = cls(config, is_gqa=self.is_gqa)
            layer.cross_attention.cache = cls(config, is_gqa=self.is_gqa)

    def _clean_cache(self) -> None:
        for layer in self.decoder.all_layer:
            layer.attention.cache = None
            layer.cross_attention.cache = None

    @classmethod
    def from_config(cls, encoder_config, decoder_config, encoder: Optional[nn.Module] = None,
                    encoder_pos_embedding_type: Optional[str] = "absolute",
                    encoder_attention_type: Optional[str] = None,
                    decoder_pos_embedding_type: Optional[str] = "absolute",
                    decoder_attention_type: Optional[str] = None) -> nn.Module:
        return cls(encoder_config, decoder_config, encoder, encoder_pos_embedding_type, encoder_attention_type,
                   decoder_pos_embedding_type, decoder_attention_type)
