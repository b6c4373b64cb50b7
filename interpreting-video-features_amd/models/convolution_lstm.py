"""Drop-in for video_features_pytorch/models/convolution_lstm.py: parameter
containers with the reference's names and creation order; the arithmetic runs in
csrc/convlstm.hip through models.CLSTM_4.Model."""
import torch
import torch.nn as nn

import ivf_lib as L


class ConvLSTMCell(nn.Module):
    """convolution_lstm.py:10-36 (8 Conv2d holders; peepholes are zero non-parameters)."""

    def __init__(self, input_channels, hidden_channels, kernel_size, conv_stride, device='cpu'):
        super().__init__()
        assert hidden_channels % 2 == 0
        self.input_channels, self.hidden_channels = input_channels, hidden_channels
        self.kernel_size, self.conv_stride, self.device = kernel_size, conv_stride, device
        self.num_features = 4
        self.padding = int((kernel_size - 1) / 2)
        for g in "ifco":
            setattr(self, "Wx" + g, nn.Conv2d(input_channels, hidden_channels, kernel_size, conv_stride,
                                              self.padding, bias=True))
            setattr(self, "Wh" + g, nn.Conv2d(hidden_channels, hidden_channels, kernel_size, 1, self.padding,
                                              bias=False))
        self.Wci = self.Wcf = self.Wco = None

    def forward(self, x, h, c):
        raise L.IvfError("ConvLSTMCell is executed inside the HIP plan; call models.CLSTM_4.Model")


class ConvLSTM(nn.Module):
    """convolution_lstm.py:63-94."""

    def __init__(self, input_channels, hidden_channels, kernel_size, conv_stride, pool_kernel_size=(2, 2), step=1,
                 effective_step=[1], batch_normalization=True, dropout=0, device='cpu'):
        super().__init__()
        self.input_channels = [input_channels] + hidden_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.num_layers = len(hidden_channels)
        self.step, self.effective_step = step, effective_step
        self.pool_kernel_size, self.conv_stride = pool_kernel_size, conv_stride
        self.mp = nn.MaxPool2d(kernel_size=self.pool_kernel_size)
        self.batch_norm = batch_normalization
        self.dropout_rate = dropout
        self.device = device
        self.bn = nn.BatchNorm2d(self.hidden_channels[0], eps=1e-05, momentum=0.1, affine=True)
        self.dropout = torch.nn.Dropout(p=self.dropout_rate)
        self._all_layers = []
        for i in range(self.num_layers):
            cell = ConvLSTMCell(self.input_channels[i], self.hidden_channels[i], self.kernel_size, self.conv_stride,
                                self.device)
            setattr(self, 'cell{}'.format(i), cell)
            self._all_layers.append(cell)

    def forward(self, input):
        raise L.IvfError("ConvLSTM is executed inside the HIP plan; call models.CLSTM_4.Model")
