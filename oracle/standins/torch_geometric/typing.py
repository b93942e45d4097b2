"""Type aliases the reference imports from torch_geometric.typing."""
from typing import Optional, Tuple, Union
from torch import Tensor

Adj = Tensor
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
OptPairTensor = Tuple[Tensor, Optional[Tensor]]
Size = Optional[Tuple[int, int]]
NoneType = Optional[Tensor]
