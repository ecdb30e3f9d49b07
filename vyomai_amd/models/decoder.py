"""GPT-style decoder with the reference's API (VyomAI/models/decoder.py): DecoderAttention /
DecoderAttentionGqa that thread ``kv_cache`` through forward, DecoderLayer, LMHead, DecoderModel
with create_mask_for_decoder and the batched greedy ``generate``.  All layer math runs in the
HIP kernels; this file is host-side orchestration."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import VyomHipError
from ..layers.attention import _SelfAttentionBase
from ..layers.ffn import FeedForward
from ..layers.kv_cache import DynamicCacheOne, StaticCacheOne
from ..layers.mask import AttnMask
from ..autograd_train import defer_residual_grads as _defer
from .common import LMHead, PositionMixin


@dataclass
class DecoderOutput(object):
    logits: torch.Tensor


@dataclass
class CLMOutput(object):
    hidden_state: torch.Tensor
    logits: torch.Tensor
    kv_cache: List[torch.FloatTensor] = None


class DecoderAttention(_SelfAttentionBase):
    """forward(hidden_state, attention_mask, freqs, use_cache, kv_cache, start_pos)
    -> (hidden_state, kv_cache).  Reference :44-113."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, None, fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                kv_cache=None, start_pos: Optional[int] = 0):
        if use_cache and kv_cache is None:
            raise ValueError("you need to pass kv_cache")
        out = self._attend(hidden_state, attention_mask, freqs, kv_cache if use_cache else None,
                           self.layer_idx, start_pos)
        return out, kv_cache


class DecoderAttentionGqa(_SelfAttentionBase):
    """Reference :116-201 (K/V projections of num_key_value_heads heads, default 4)."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, getattr(config, "num_key_value_heads", 4), fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                kv_cache=None, start_pos: Optional[int] = 0):
        if use_cache and kv_cache is None:
            raise ValueError("you need to pass kv_cache")
        out = self._attend(hidden_state, attention_mask, freqs, kv_cache if use_cache else None,
                           self.layer_idx, start_pos)
        return out, kv_cache


class DecoderLayer(nn.Module):
    """a = attention(h); return feed_forward(a, h): the FFN residual is the LAYER INPUT, as in the
    reference (:241-250)."""

    def __init__(self, config, layer_idx: int, attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.attention = (DecoderAttentionGqa(config, layer_idx=layer_idx) if attention_type == "gqa"
                          else DecoderAttention(config, layer_idx=layer_idx))
        if attention_type == "gqa" and layer_idx == 0:
            print("Decoder Using GQA Attention")
        self.feed_forward = FeedForward(config)
        self.layer_idx = layer_idx

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                kv_cache=None, start_pos: Optional[int] = 0):
        _defer(hidden_state)  # training: its residual-path gradients are added in the QKV dgrad epilogue
        out, kv_cache = self.attention(hidden_state=hidden_state, attention_mask=attention_mask, freqs=freqs,
                                       use_cache=use_cache, kv_cache=kv_cache, start_pos=start_pos)
        return self.feed_forward(out, hidden_state), kv_cache


def _two_lanes(model, rows: int) -> bool:
    """Run the forward as two batch halves on two HIP streams (ops.lanes)?  Opt-in (VY_LANES=1): on configs[1] the two
    half-size launch chains do overlap (two or more kernels resident 65-72 % of the time) but every half-size GEMM then
    takes as long as the full-size one did -- forward 6.63 against 6.49 ms, training step 18.71 against 18.56 ms, same box,
    interleaved (tools/ab_env.py VY_LANES 0 1)."""
    return os.environ.get("VY_LANES", "0") == "1"


class DecoderModel(nn.Module, PositionMixin):
    """Reference :278-514."""

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
        self.config = config

    def forward_hidden(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                       use_cache: Optional[bool] = False, kv_cache=None, start_pos: Optional[int] = 0):
        """Everything of forward() up to the LM head -> (hidden_state, kv_cache)."""
        _bsz, seqlen = input_ids.shape
        hidden_state = self._embed(self.word_embeddings, input_ids)
        hidden_state, freqs = self._positions(hidden_state, start_pos, seqlen)
        mask = None
        if seqlen > 1:
            mask = self.create_mask_for_decoder(input_ids=input_ids, attention_mask=attention_mask,
                                                start_pos=start_pos)
        for layer in self.all_layer:
            hidden_state, kv_cache = layer(hidden_state, mask, freqs=freqs, use_cache=use_cache,
                                           kv_cache=kv_cache, start_pos=start_pos)
        return hidden_state, kv_cache

    def clm_loss(self, input_ids: torch.Tensor, labels: torch.Tensor,
                 attention_mask: Optional[torch.Tensor] = None, ignore_index: int = -100) -> torch.Tensor:
        """Shifted next-token loss (Examples/vyom-ai-decoder_clm.ipynb cell 29) with the LM head and
        the cross-entropy fused (no logits copy)."""
        B, L = input_ids.shape
        with ops.lanes(B, L, _two_lanes(self, B * L)):
            hidden_state, _ = self.forward_hidden(input_ids, attention_mask)
            return self.lm_head.loss(hidden_state, labels, ignore_index)

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = False, kv_cache=None, start_pos: Optional[int] = 0) -> CLMOutput:
        B, L = input_ids.shape
        with ops.lanes(B, L, not use_cache and _two_lanes(self, B * L)):
            hidden_state, kv_cache = self.forward_hidden(input_ids, attention_mask, use_cache, kv_cache, start_pos)
            logits = self.lm_head(hidden_state)
        return CLMOutput(hidden_state=hidden_state, logits=logits, kv_cache=kv_cache)

    def create_mask_for_decoder(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                                start_pos: Optional[int] = 0) -> AttnMask:
        """The reference (:376-419) returns the dense (B,1,L,start+L) 0/1 product of the causal
        and padding masks, which forward() inverts to an additive tensor.  Here the same
        information is returned as a descriptor the attention kernel evaluates in registers;
        ``.dense()`` rebuilds the reference's additive tensor."""
        _, seq_length = input_ids.shape
        return AttnMask.from_padding(attention_mask, causal=True, start_pos=start_pos, query_len=seq_length)

    @classmethod
    def from_config(cls, config, pos_embedding_type: Optional[str] = "absolute",
                    attention_type: Optional[str] = None) -> nn.Module:
        return cls(config, pos_embedding_type, attention_type)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, max_len: int = 5,
                 temperature: float = 1.0, use_cache: bool = True, do_sample: bool = False,
                 use_static_cache: bool = False) -> torch.Tensor:
        """Batched generation loop of the reference (:430-514): fills a (B, prompt+max_len) token
        matrix, forcing prompt tokens while a row is still inside its prompt; greedy = top-1 of the
        last position.  The end-of-sequence bookkeeping stays on the device and is checked with
        one scalar read per token."""
        device = input_ids.device
        if device.type != "cuda":
            raise VyomHipError("vyomai_amd ops run on MI355X only: generate() got CPU token ids (no CPU fallback "
                               "exists; move the model and its inputs to 'cuda')")
        bsz, prompt_len = input_ids.shape
        total = max_len + prompt_len
        pad_id = getattr(self.config, "pad_token_id", 1)
        tokens = torch.full((bsz, total), pad_id, dtype=torch.long, device=device)
        tokens[:, :prompt_len] = input_ids
        kv_cache = None
        if use_cache:
            if use_static_cache:
                kv_cache = StaticCacheOne(self.config, max_cache_len=total, batch_size=bsz,
                                          dtype=self.compute_dtype or self.word_embeddings.weight.dtype)
            else:
                kv_cache = DynamicCacheOne(self.config)
        plan = None  # native single-token driver (one C call per step) once the prompt is cached
        prev_pos = 0
        eos_reached = torch.zeros(bsz, dtype=torch.bool, device=device)
        input_text_mask = tokens != pad_id
        stop_tokens = torch.tensor(getattr(self.config, "eos_token_id", 2), device=device)
        # greedy: one kernel per position does the top-1, the prompt forcing, the token write and the EOS
        # bookkeeping (vy_greedy_step) and counts the rows still running; the host reads that count one
        # step late from pinned memory, so a step is always queued behind the running one.
        greedy = not do_sample
        if greedy:
            stop_ids = stop_tokens.reshape(-1).to(torch.long).contiguous()
            not_done = torch.zeros(total, dtype=torch.int32, device=device)
            done_host = torch.empty(total, dtype=torch.int32).pin_memory()
            landed = {}
        for cur_pos in range(prompt_len, total):
            if greedy and cur_pos - 2 in landed:
                landed.pop(cur_pos - 2).synchronize()
                if int(done_host[cur_pos - 2]) == 0:
                    # every row had reached EOS after position cur_pos-2: the reference stops there (:512-513);
                    # position cur_pos-1 was generated in the meantime and goes back to padding
                    tokens[:, cur_pos - 1] = pad_id
                    return tokens
            if use_cache and use_static_cache and cur_pos - prev_pos == 1 and device.type == "cuda":
                if plan is None:
                    from ..decode_plan import DecodePlan
                    plan = DecodePlan(self, kv_cache, bsz, self.compute_dtype or self.word_embeddings.weight.dtype,
                                      device)
                hidden = self._embed(self.word_embeddings, tokens[:, prev_pos:cur_pos])
                hidden, _ = self._positions(hidden, prev_pos, 1)
                next_token_logits, _ = plan.step(hidden[:, 0, :].contiguous(), prev_pos)
            else:
                # only the last position's logits are used (reference :478-489 slices them out of the full
                # (B, L, V) tensor): the vocabulary projection runs on that one row per sequence
                hidden, kv_cache = self.forward_hidden(tokens[:, prev_pos:cur_pos], attention_mask, use_cache,
                                                       kv_cache, prev_pos)
                next_token_logits = self.lm_head(hidden[:, -1:, :].contiguous())[:, -1]
            if greedy:
                # (the division by the temperature does not move the maximum)
                ops.greedy_step_(next_token_logits, tokens, cur_pos, input_text_mask, stop_ids, eos_reached,
                                 not_done[cur_pos:cur_pos + 1])
                done_host[cur_pos:cur_pos + 1].copy_(not_done[cur_pos:cur_pos + 1], non_blocking=True)
                landed[cur_pos] = torch.cuda.Event()
                landed[cur_pos].record()
            else:
                # the reference samples from the raw logits (:491-492); kept
                next_token = torch.multinomial((next_token_logits / temperature).float(), num_samples=1).reshape(-1)
                next_token = torch.where(input_text_mask[:, cur_pos], tokens[:, cur_pos], next_token)
                tokens[:, cur_pos] = next_token
                eos_reached |= (~input_text_mask[:, cur_pos]) & torch.isin(next_token, stop_tokens)
            if use_cache:
                prev_pos = cur_pos
            if plan is None:
                attention_mask = torch.cat(
                    [attention_mask, torch.ones((bsz, 1), device=device, dtype=attention_mask.dtype)], dim=-1)
            if not greedy and bool(eos_reached.all()):
                break
        if greedy:
            # the last two positions have not been looked at yet
            torch.cuda.current_stream().synchronize()
            for pos in sorted(landed):
                if int(done_host[pos]) == 0:
                    tokens[:, pos + 1:] = pad_id
                    break
        return tokens
