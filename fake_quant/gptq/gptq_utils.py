"""GPTQ (Frantar et al., arXiv:2210.17323) for nn.Linear: Hessian accumulation over calibration
inputs, damped Cholesky inverse, column-blocked quantization with error feedback, optional
activation ordering.  Host-side torch code (offline, one-shot); the reference's counterpart is
``fake_quant/gptq/gptq_utils.py:171-310``."""
import logging
import math

import torch

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False


class GPTQ:
    #: CUDA weights + symmetric per-channel quantizer without groups: column loop in one HIP launch
    use_kernel = True

    def __init__(self, layer):
        self.layer = layer
        self.dev = layer.weight.device
        self.rows, self.columns = layer.weight.shape[0], layer.weight.data.flatten(1).shape[1]
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self.quantizer = None

    @staticmethod
    def _kernel_device(W):
        return W.is_cuda

    @staticmethod
    def _block(W, i1, i2, Hrows, scale, bits, Q, E1):
        """Column loop of one block: reads W[:, i1:i2], writes Q[:, i1:i2] and E1 (mq_gptq_block)."""
        from mquant_amd import ops
        ops.gptq_block(W, i1, i2, Hrows, scale, bits, Q, E1)

    def add_batch(self, inp, out=None):
        """Running mean of 2 x x^T over every token seen so far."""
        if isinstance(self.layer, (torch.nn.Conv2d, torch.nn.Conv3d)):
            # patch embedding: kernel == stride == the whole patch, so unfolding is a reshape
            # (reference GPTQConv.add_batch, gptq_utils.py:49-60, through nn.Unfold / UnfoldNd)
            assert tuple(inp.shape[2:]) == tuple(self.layer.kernel_size), "only whole-patch convolutions"
            batch = inp.shape[0]
            x = inp.reshape(batch, -1).t().float()
        else:
            if inp.dim() == 2:
                inp = inp.unsqueeze(0)
            batch = inp.shape[0]
            x = inp.reshape(-1, inp.shape[-1]).t().float()
        self.H *= self.nsamples / (self.nsamples + batch)
        self.nsamples += batch
        x = math.sqrt(2 / self.nsamples) * x
        self.H += x @ x.t()

    def fasterquant(self, blocksize=128, percdamp=0.01, groupsize=-1, actorder=False,
                    static_groups=False):
        W = self.layer.weight.data.clone().flatten(1).float()
        if not self.quantizer.ready():
            self.quantizer.find_params(W)
        H = self.H
        self.H = None
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        W[:, dead] = 0
        groups = None
        if static_groups:
            import copy
            groups = []
            for i in range(0, self.columns, groupsize):
                g = copy.deepcopy(self.quantizer)
                g.find_params(W[:, i:i + groupsize])
                groups.append(g)
        perm = invperm = None
        if actorder:
            perm = torch.argsort(torch.diag(H), descending=True)
            W, H = W[:, perm], H[perm][:, perm]
            invperm = torch.argsort(perm)
        Q = torch.zeros_like(W)
        idx = torch.arange(self.columns, device=self.dev)
        H[idx, idx] += percdamp * torch.mean(torch.diag(H))
        Hinv = torch.cholesky_inverse(torch.linalg.cholesky(H))   # a non-positive-definite Hessian raises, as upstream
        try:
            Hinv = torch.linalg.cholesky(Hinv, upper=True)
        except Exception:   # upstream guards only this last factorisation: fall back to plain RTN
            logging.warning("GPTQ: upper Cholesky of the inverse Hessian failed, falling back to RTN")
            Wq = self.quantizer.quantize(W if perm is None else W[:, invperm])
            self.layer.weight.data = Wq.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
            return
        qz = self.quantizer
        fused = (self.use_kernel and self._kernel_device(W) and groupsize == -1 and blocksize <= 128 and qz.sym
                 and getattr(qz, "perchannel", False) and 2 <= qz.bits <= 8)
        if fused:
            # the per-column loop as ONE launch per block (mq_gptq_block, same operation order)
            scale = qz.scale.reshape(-1).to(device=W.device, dtype=torch.float32).contiguous()
            W = W.contiguous()
            Hrows = Hinv.contiguous()      # row-major copy for the kernel; the trailing GEMM keeps
            #                                upstream's operand layout (the upper factor comes back
            #                                transposed) so that its summation order is unchanged
            for i1 in range(0, self.columns, blocksize):
                i2 = min(i1 + blocksize, self.columns)
                E1 = torch.empty((self.rows, i2 - i1), dtype=torch.float32, device=W.device)
                self._block(W, i1, i2, Hrows, scale, qz.bits, Q, E1)
                W[:, i2:] -= E1 @ Hinv[i1:i2, i2:]
        # --w_groupsize: every group's (scale, zero point) is KEPT (the reference's quantizer remembers the last group only,
        # gptq_utils.py:263-273); with contiguous groups -- no activation ordering -- they are what the integer backend needs
        # (mq_gemm_w4a8_wgroupscale): quantizer.group_scales / group_zeros [rows, groups], quantizer.groupsize
        group_scales, group_zeros = [], []
        for i1 in ([] if fused else range(0, self.columns, blocksize)):
            i2 = min(i1 + blocksize, self.columns)
            W1 = W[:, i1:i2].clone()
            Q1 = torch.zeros_like(W1)
            E1 = torch.zeros_like(W1)
            Hb = Hinv[i1:i2, i1:i2]
            for i in range(i2 - i1):
                w, d = W1[:, i], Hb[i, i]
                if groupsize != -1:
                    if not static_groups:
                        if (i1 + i) % groupsize == 0:
                            self.quantizer.find_params(W[:, (i1 + i):(i1 + i + groupsize)])
                            group_scales.append(self.quantizer.scale.reshape(-1).float().clone())
                            group_zeros.append(self.quantizer.zero.reshape(-1).float().clone())
                    else:
                        col = i1 + i
                        self.quantizer = groups[(int(perm[col]) if actorder else col) // groupsize]
                q = self.quantizer.quantize(w.unsqueeze(1)).flatten()
                Q1[:, i] = q
                err = (w - q) / d
                W1[:, i:] -= err.unsqueeze(1) @ Hb[i, i:].unsqueeze(0)
                E1[:, i] = err
            Q[:, i1:i2] = Q1
            W[:, i2:] -= E1 @ Hinv[i1:i2, i2:]
        if groupsize != -1 and not static_groups:
            qz.groupsize = int(groupsize)
            # --act_order: the groups are runs of PERMUTED columns (column j of the solver's order is column perm[j] of the weight);
            # the permutation is kept with the scales, and the integer backend gathers the activation columns the same way
            qz.group_permuted = bool(actorder)
            qz.group_perm = perm.clone() if actorder else None
            qz.group_scales = torch.stack(group_scales, dim=1)
            qz.group_zeros = torch.stack(group_zeros, dim=1)
        if actorder:
            Q = Q[:, invperm]
        self.layer.weight.data = Q.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
        if torch.any(torch.isnan(self.layer.weight.data)):
            logging.warning("NaN in weights")
            raise ValueError("NaN in weights")

    def free(self):
        self.H = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
