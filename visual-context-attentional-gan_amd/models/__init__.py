"""Drop-in mirrors of the reference's src/models modules (same class names, constructor arguments, forward
signatures and state_dict keys), computing on the HIP kernels of libvca_hip.so."""
from .generator import (AVAttention, Decoder, Discriminator, GenResBlk, Postnet, ResBlk, ResBlk1D,  # noqa: F401
                        final_length, gan_loss, sync_Discriminator)
from .resnet import BasicBlock, ResNet  # noqa: F401
from .visual_front import Visual_front  # noqa: F401
