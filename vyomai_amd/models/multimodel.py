"""Vision-language fusion model with the reference's API (VyomAI/models/multimodel.py): the ViT
CLS vector is prepended as token 0 of the decoder; per-layer KV caches are attached to the
attention modules by _setup_cache()."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from ..autograd_train import defer_residual_grads as _defer
from ..layers.attention import DecoderAttention, DecoderAttentionGqa
from ..layers.ffn import FeedForward
from ..layers.kv_cache import DynamicCache, StaticCache
from ..layers.mask import AttnMask
from .common import LMHead, PositionMixin


@dataclass
class DecoderOutput(object):
    logits: torch.Tensor


class DecoderLayer(nn.Module):
    """Reference :26-69 (cache lives on ``self.attention.cache``)."""

    def __init__(self, config, layer_idx: int, attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.attention = (DecoderAttentionGqa(config, layer_idx=layer_idx) if attention_type == "gqa"
                          else DecoderAttention(config, layer_idx=layer_idx))
        if attention_type == "gqa" and layer_idx == 0:
            print("Decoder Using GQA Attention")
        self.feed_forward = FeedForward(config)
        self.layer_idx = layer_idx

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> torch.Tensor:
        _defer(hidden_state)  # training: its residual-path gradients are added in the QKV dgrad epilogue
        out = self.attention(hidden_state=hidden_state, attention_mask=attention_mask, freqs=freqs,
                             use_cache=use_cache, start_pos=start_pos)
        return self.feed_forward(out, hidden_state)


class VisionLanguageDecoderModel(nn.Module, PositionMixin):
    """Reference :97-255."""

    def __init__(self, config, pos_embedding_type: Optional[str] = "absolute",
                 attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.is_gqa = attention_type == "gqa"
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                            padding_idx=getattr(config, "pad_token_id", None))
        self._init_positions(config, pos_embedding_type, "Decoder")
        self.all_layer = nn.ModuleList(
            [DecoderLayer(config, i, attention_type) for i in range(config.num_hidden_layers)])
        self.lm_head = LMHead(config=config)

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                encoder_hidden_state: Optional[torch.Tensor] = None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> DecoderOutput:
        bsz, _ = input_ids.shape
        hidden_state = self._embed(self.word_embeddings, input_ids)
        if start_pos == 0:  # the image vector is token 0 of the sequence (:163-169)
            hidden_state = torch.cat([encoder_hidden_state.to(hidden_state.dtype).unsqueeze(1), hidden_state], dim=1)
            if attention_mask is not None:
                one = torch.ones((bsz, 1), dtype=attention_mask.dtype, device=attention_mask.device)
                attention_mask = torch.cat([one, attention_mask], dim=1)
        seqlen = hidden_state.shape[1]
        hidden_state, freqs = self._positions(hidden_state, start_pos, seqlen)
        mask = None
        if seqlen > 1:
            mask = self.create_mask_for_decoder(hidden_state=hidden_state, attention_mask=attention_mask,
                                                start_pos=start_pos)
        for layer in self.all_layer:
            hidden_state = layer(hidden_state, mask, freqs=freqs, use_cache=use_cache, start_pos=start_pos)
        if getattr(self, "_want_hidden", False):
            return hidden_state
        return DecoderOutput(logits=self.lm_head(hidden_state))

    def create_mask_for_decoder(self, hidden_state: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                                start_pos: Optional[int] = 0) -> AttnMask:
        """Descriptor form of the reference's dense causal x padding mask (:203-246)."""
        return AttnMask.from_padding(attention_mask, causal=True, start_pos=start_pos,
                                     query_len=hidden_state.shape[1])

    @classmethod
    def from_config(cls, config, pos_embedding_type: Optional[str] = "absolute",
                    attention_type: Optional[str] = None) -> nn.Module:
        return cls(config, pos_embedding_type, attention_type)


class VisionLanguageModel(nn.Module):
    """Reference :258-314."""

    def __init__(self, config, encoder, pos_embedding_type: Optional[str] = "absolute",
                 attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.is_gqa = attention_type == "gqa"
        self.encoder = encoder
        self.decoder = VisionLanguageDecoderModel(config=config, pos_embedding_type=pos_embedding_type,
                                                  attention_type=attention_type)

    def forward(self, pixel_values=None, decoder_input_ids=None, decoder_attention_mask=None,
                encoder_output=None, use_cache: Optional[bool] = False, start_pos: Optional[int] = 0) -> DecoderOutput:
        if encoder_output is None:
            encoder_output = self.encoder(pixel_values=pixel_values).logits[:, 0, :]
        return self.decoder(input_ids=decoder_input_ids, attention_mask=decoder_attention_mask,
                            encoder_hidden_state=encoder_output, use_cache=use_cache, start_pos=start_pos)

    def caption_loss(self, pixel_values, decoder_input_ids, decoder_attention_mask=None,
                     ignore_index: int = -100) -> torch.Tensor:
        """Next-token loss of the caption with the LM head and the cross-entropy fused (no logits copy), the
        training-side entry point like DecoderModel.clm_loss.  The decoder sees [image, t0 .. tn-1]; position p
        predicts position p + 1, and the first caption token t0 (predicted from the image token alone) is not
        scored -- the loss of the reference's captioning notebooks: cross_entropy(logits[:, 1:-1], ids[:, 1:])."""
        self.decoder._want_hidden = True
        try:
            hidden = self.forward(pixel_values=pixel_values, decoder_input_ids=decoder_input_ids,
                                  decoder_attention_mask=decoder_attention_mask)
        finally:
            self.decoder._want_hidden = False
        B = decoder_input_ids.shape[0]
        pad = torch.full((B, 2), ignore_index, dtype=decoder_input_ids.dtype, device=decoder_input_ids.device)
        labels = torch.cat([pad, decoder_input_ids[:, 1:]], dim=1)      # aligned with [image, t0, t1, ...]
        if decoder_attention_mask is not None:
            labels[:, 2:] = labels[:, 2:].masked_fill(decoder_attention_mask[:, 1:] == 0, ignore_index)
        return self.decoder.lm_head.loss(hidden, labels, ignore_index)

    def get_decoder(self) -> nn.Module:
        return self.decoder

    def get_encoder_output(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self.encoder(pixel_values=pixel_values).logits[:, 0, :]

    def _setup_cache(self, config, cls: Optional[object] = StaticCache) -> None:
        for layer in self.decoder.all_layer:
            layer.attention.cache = cls(config, is_gqa=self.is_gqa)

    def _clean_cache(self) -> None:
        for layer in self.decoder.all_layer:
            layer.attention.cache = None
