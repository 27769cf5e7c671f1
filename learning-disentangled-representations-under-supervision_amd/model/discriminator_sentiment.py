"""Text-CNN sentiment discriminator (Kim 2014 style). Same class, attribute names (convs1, dropout, fc1) and forward
signature as the reference's model/discriminator_sentiment.py:11-78. Convolutions are implicit-GEMM MFMA kernels with
a fused ReLU + max-over-time; the two-class head is one fused kernel."""
import torch
import torch.nn as nn

from ..hip import nn as hnn


class Sentiment_CNN(nn.Module):

    def __init__(self, config):
        super(Sentiment_CNN, self).__init__()
        self.config = config

        D = config.word_embed_size
        C = 2   # Positive/Negative
        Co = config.sentiment_kernel_num
        Ks = config.sentiment_kernel_size

        self.convs1 = nn.ModuleList([hnn.Conv2dParams(Co, K, D) for K in Ks])
        self.dropout = nn.Dropout(config.sentiment_dropout)   # holds p; the mask multiply is fused into the head
        self.fc1 = hnn.Linear(len(Ks) * Co, C)

    def forward(self, x, drop_mask=None):
        """
        x should be (batch_size, seq_len, D). Returns logits (batch_size, 2).
        `drop_mask` (optional, [batch_size, len(Ks)*Co], already scaled by 1/(1-p)) overrides the random dropout mask
        of discriminator_sentiment.py:76; without it a mask is drawn with torch's RNG in train() mode.
        """
        if x.size(1) < max(self.config.sentiment_kernel_size):
            # nn.Conv2d raises the same way in the reference
            raise RuntimeError("Kernel size can't be greater than actual input size: seq_len {} < {}".format(
                x.size(1), max(self.config.sentiment_kernel_size)))
        if drop_mask is None and self.training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            F_ = len(self.convs1) * self.config.sentiment_kernel_num
            drop_mask = (torch.rand(x.size(0), F_, device=x.device) < keep).to(torch.float32) / keep
        flat = []
        for conv in self.convs1:
            flat += [conv.weight, conv.bias]
        return hnn._TextCNNFn.apply(x, drop_mask, self.fc1.weight, self.fc1.bias, *flat)
